#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DDCTN_STAMPS; never the shipped one): phase stamps of the ConvSBS backward
kernel's first window group per workgroup.   make -C dctn_amd/csrc CXXFLAGS+=-DDCTN_STAMPS ; python tools/stamp_sbs.py 16"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dctn_amd import _lib as L
from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSSpecCore
from dctn_amd.pos2d import Pos2D

r = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SNAKE = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
dev = torch.device("cuda:0")
spec = (tuple(SBSSpecCore(Pos2D(*p), 2 if i == 4 else 1) for i, p in enumerate(SNAKE)),)
many = ManyConvSBS(1, 3, r, False, spec, (DumbNormalInitialization((3 * r) ** -0.5),)).to(dev)
x = torch.randn(1, 128, 32, 32, 3, device=dev, requires_grad=True)
for _ in range(3):
    x.grad = None
    (y,) = many(x)
    y.backward(torch.randn_like(y))
torch.cuda.synchronize()
lib = ctypes.CDLL(L.LIB_PATH)
n = 2048 * 32
buf = (ctypes.c_ulonglong * n)()
rc = lib.dctn_debug_read_sbs_stamps(buf, n)
st = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 32).astype(np.int64)
ok = st[:, 0] > 0
st = st[ok]
t0 = st[:, 0].min()
names = {0: "entry", 1: "cores packed", 2: "features staged", 3: "forward sweep done", 4: "last core done", 20: "first core done (group end)",
         21: "all groups done", 22: "register tiles joined", 23: "flushed (end)"}
for c in range(7):
    names[5 + c] = f"adjoint core {7 - c} done"
print(f"r={r}: {ok.sum()} workgroups; us after the first entry (min / median / max)")
for slot in sorted(names):
    v = st[:, slot]
    v = (v[v > 0] - t0) * 0.01
    if len(v):
        print(f"  {names[slot]:32s} {v.min():8.2f} {np.median(v):8.2f} {v.max():8.2f}")

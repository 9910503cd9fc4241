#!/usr/bin/env python3
"""Diagnostic: forward / forward+backward time of a 9-core snake ConvSBS as open chain and as ring (trace_edge).
   python tools/time_ring.py [bond]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dctn_amd import _lib as L
from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSSpecCore
from dctn_amd.pos2d import Pos2D
r = int(sys.argv[1]) if len(sys.argv) > 1 else 4
SNAKE = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
dev = torch.device("cuda:0")
spec = (tuple(SBSSpecCore(Pos2D(*p), 2 if i == 4 else 1) for i, p in enumerate(SNAKE)),)
x = torch.rand(1, 128, 28, 28, 2, device=dev, requires_grad=True)
for ring in (False, True):
    many = ManyConvSBS(1, 2, r, ring, spec, (DumbNormalInitialization((2 * r) ** -0.5),)).to(dev)
    def fb():
        x.grad = None
        (y,) = many(x)
        y.backward(torch.ones_like(y))
    def f():
        with torch.no_grad():
            many(x)
    for fn, name in ((f, "fwd"), (fb, "fwd+bwd")):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        print(f"bond {r} ring={ring} {name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms   ({L.last_kernel()})")

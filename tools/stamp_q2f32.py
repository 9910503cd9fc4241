#!/usr/bin/env python3
"""Diagnostic: the life of every wave of the float32 register family's forward kernel - shader-clock stamps at the start,
after every step, around the head's barrier.  Loads tools/libdctn_amd.stamps.so = the library with eps_q2f32.hip compiled
with -DDCTN_STAMPS (never the shipped library):
    cd dctn_amd/csrc && hipcc <CXXFLAGS of the Makefile> -DDCTN_STAMPS -c eps_q2f32.hip -o /tmp/q.o && \
        hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libdctn_amd.stamps.so /tmp/q.o <the other objects>"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from dctn_amd import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdctn_amd.stamps.so")   # the diagnostic build
from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
model = EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, dev, torch.float32, image_size=28)
x = bench.synthetic_input(B, 28, 2, torch.float32, dev, 1)
core = model.epses[0].detach().contiguous()
w, b = model.linear.weight.detach().contiguous(), model.linear.bias.detach().contiguous()
t = bench.eps_call_timers(core, x, False, dev, head=(w, b))
for _ in range(5):
    t["head_fwd"]()
torch.cuda.synchronize()
n = 2048 * 16
buf = (ctypes.c_ulonglong * n)()
lib = L.lib()
lib.dctn_debug_read_qf_stamps.restype = ctypes.c_int
assert lib.dctn_debug_read_qf_stamps(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 16).astype(np.int64)
a = a[a[:, 0] > 0]
hw = a[:, 15]
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
print("waves with stamps:", len(a))
wg = a.reshape(-1, 8, 16)
print("SIMD ids of the 8 waves of the first workgroups:", [list(((wg[i, :, 15] >> 4) & 3)) for i in range(4)])
t0 = wg[:, :, 0].min(axis=1, keepdims=True)
rel = wg[:, :, :13] - t0[:, :, None]
np.set_printoptions(linewidth=200)
print("median over workgroups, per wave: start, end of steps 1..6, before barrier, after barrier, end of head (cycles from the workgroup's first stamp)")
for wv in range(8):
    r = np.median(rel[:, wv, :], axis=0)
    steps = [int(v) for v in r[1:10]]
    print(f"  wave {wv}: start {int(r[0]):6d}  steps {steps}  barrier {int(r[10])} -> {int(r[11])}  end {int(r[12])}")
d = np.diff(rel[:, :, 1:7], axis=2)
print("step durations (cycles), median over all waves:", [int(v) for v in np.median(d.reshape(-1, d.shape[2]), axis=0)])
print("kernel span (max end - min start) median over workgroups:", int(np.median(rel[:, :, 12].max(axis=1))))

#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DDCTN_STAMPS: `make -C dctn_amd/csrc clean && make -C dctn_amd/csrc EXTRA=-DDCTN_STAMPS`;
never the shipped one): where wave 0 of every workgroup of the large-core forward kernel spends its cycles.
    python tools/stamp_bigcore.py [B] [fwd|dx|dcore] [K Q O H]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch   # noqa: E402

import dctn_amd   # noqa: E402
from dctn_amd import _lib as L   # noqa: E402
from dctn_amd.eps import eps   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
WHAT = sys.argv[2] if len(sys.argv) > 2 else "fwd"   # "fwd": eps_bigcore_k forward; "dx": its transposed (dX) launches; "dcore": eps_bigcore_dcore_k
dev = torch.device("cuda")
# layer shape: K Q O H (default: cfg3a layer 2 - 3 4 6 on 25 x 25; cfg3b layer 1 is 4 2 8 28)
K, Q, O, H = (int(v) for v in sys.argv[3:7]) if len(sys.argv) > 6 else (3, 4, 6, 25)
core = torch.randn(*(Q,) * (K * K), O, device=dev) * Q ** (-K * K / 4)
x = torch.rand(1, B, H, H, Q, device=dev) + 0.1
n = 16384 * 8
buf = (ctypes.c_ulonglong * n)()
lib = L.lib()
if WHAT == "fwd":
    with torch.no_grad():
        for _ in range(3):
            y = eps(core, x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = eps(core, x)
        e1.record()
    torch.cuda.synchronize()
    call_ms = e0.elapsed_time(e1)
    lib.dctn_debug_read_bc_stamps.restype = ctypes.c_int
    rc = lib.dctn_debug_read_bc_stamps(buf, n)
    names = ["total", "prologue", "waiting at the stage barrier", "core-tile fetch issue", "operand + MFMA blocks", "core-tile commit",
             "row-tile epilogues", "first-stage fetch + commit"]
elif WHAT == "dx":
    x.requires_grad_(True)
    for _ in range(2):
        y = eps(core, x)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    lib.dctn_debug_read_bc_stamps_g.restype = ctypes.c_int
    rc = lib.dctn_debug_read_bc_stamps_g(buf, n)
    names = ["total", "prologue", "waiting at the stage barrier", "core-tile fetch issue", "operand + MFMA blocks", "core-tile commit",
             "row-tile epilogues", "first-stage fetch + commit"]
else:
    core.requires_grad_(True)
    for _ in range(2):
        y = eps(core, x)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    lib.dctn_debug_read_dc_stamps.restype = ctypes.c_int
    rc = lib.dctn_debug_read_dc_stamps(buf, n)
    names = ["total", "chunk commit (+ barrier before it)", "next chunk's fetch issue", "table build", "barrier after the build",
             "MFMA loop", "result store", "-"]
assert rc == 0, rc
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
a = a[a[:, 0] > 0]
print(dctn_amd.last_kernel(), "workgroups with stamps:", len(a))
tot = np.median(a[:, 0])
for i, nm in enumerate(names):
    print("  %-32s median %9.0f cycles  (%5.1f %% of total)   min %9.0f  max %9.0f" % (nm, np.median(a[:, i]), 100 * np.median(a[:, i]) / tot, a[:, i].min(), a[:, i].max()))
if WHAT == "fwd":
    # workgroups resident per CU on average: the workgroups' lifetimes over the call's duration (s_memtime ticks at 100 MHz)
    print("  call %.3f ms; sum of workgroup lifetimes / (call x 256 CUs) = %.2f workgroups per CU (ticks taken as %.0f MHz)" % (
        call_ms, a[:, 0].sum() / (call_ms * 1e-3 * 100e6 * 256), 100.0))

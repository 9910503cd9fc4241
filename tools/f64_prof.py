#!/usr/bin/env python3
"""cfg1 (float64 EPS micro-benchmark shape): forward + backward a few times (for rocprofv3 --kernel-trace
--stats), then HIP-event timings of forward and forward + backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dctn_amd.eps import eps

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.rand(1, B, 28, 28, 2, dtype=torch.float64, device=dev).requires_grad_(True)
core = (torch.randn(*(2,) * 16, 2, dtype=torch.float64, device=dev) * 2.0 ** -4).requires_grad_(True)
dy = torch.randn(B, 25, 25, 2, dtype=torch.float64, device=dev)


def fb():
    x.grad = core.grad = None
    eps(core, x).backward(dy)


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for _ in range(3):
    fb()
torch.cuda.synchronize()
with torch.no_grad():
    f = timed(lambda: eps(core, x))
print(f"variant={os.environ.get('DCTN_F64_GEMM_VARIANT', '0')} B={B} fwd_us={f:.1f} fwd_bwd_us={timed(fb):.1f}", flush=True)

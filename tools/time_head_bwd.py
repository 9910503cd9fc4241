#!/usr/bin/env python3
"""Stand-alone linear head backward (dFeat, dWeight, dBias), bf16 / f32, HIP kernels against the library GEMMs
(`eps_plus_linear.HEAD_BWD` = "hip" | "blas"): device time per call from a replayed HIP graph of 20 calls.
    python tools/time_head_bwd.py [B] [F] [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dctn_amd import eps_plus_linear as E

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2704
C = int(sys.argv[3]) if len(sys.argv) > 3 else 10
for dtype in (torch.bfloat16, torch.float32):
    f = torch.randn(B, F, device=dev).to(dtype)
    w = (torch.randn(C, F, device=dev) * F ** -0.5).to(dtype)
    g = torch.randn(B, C, device=dev).to(dtype)
    res = {}
    for mode in ("hip", "blas"):
        E.HEAD_BWD = mode
        fn = lambda: E._head_backward(f, w, g, True, True, True)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(20):
                out = fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            a.record()
            for _ in range(10):
                graph.replay()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 200 * 1e3)
        res[mode] = best
    print(f"{str(dtype):16s} B={B} F={F} C={C}: hip {res['hip']:.1f} us  blas {res['blas']:.1f} us", flush=True)

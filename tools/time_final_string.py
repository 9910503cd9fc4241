#!/usr/bin/env python3
"""Diagnostic: the layers of the reference's ConvSBS MNIST model (mnist.py:189-252) one by one: two-string layer with
one input channel, two-string layer with two input channels, final string with a 10-valued core.  python tools/time_final_string.py [bond] [ring]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dctn_amd import _lib as L
from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSSpecCore
from dctn_amd.pos2d import Pos2D
r = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ring = len(sys.argv) > 2 and sys.argv[2] == "1"
A = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
Bs = [(0, 0), (1, 0), (2, 0), (2, 1), (1, 1), (0, 1), (0, 2), (1, 2), (2, 2)]
def string(pos, mid): return tuple(SBSSpecCore(Pos2D(*p), mid if i == 4 else 1) for i, p in enumerate(pos))
dev = torch.device("cuda:0")
B = 128
cases = [("layer 1: 2 strings, 1 channel", 1, (string(A, 2), string(Bs, 2)), 28),
         ("layer 2: 2 strings, 2 channels", 2, (string(A, 2), string(Bs, 2)), 26),
         ("final: 1 string, 2 channels, 10 outputs", 2, (string(A, 10),), 24)]
for name, C, specs, size in cases:
    many = ManyConvSBS(C, 2, r, ring, specs, (DumbNormalInitialization((2 * r) ** -0.5),) * len(specs)).to(dev)
    x = torch.rand(C, B, size, size, 2, device=dev, requires_grad=True)
    def fb():
        x.grad = None
        ys = many(x)
        sum(y.sum() for y in ys).backward()
    for _ in range(3): fb()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fb()
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / 10 * 1e3
    # the same iteration replayed from a HIP graph: device time without the host's launch overhead
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fb()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fb()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print(f"bond {r} ring={ring} {name}: fwd+bwd eager {eager_ms:.3f} ms, graph replay {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms   ({L.last_kernel()})")

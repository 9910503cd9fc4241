#!/usr/bin/env python3
"""cfg3a layer 2 in bf16 (K=3, Q=4, O=6, B=128): forward + backward a few times, for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dctn_amd.eps import eps

dev = torch.device("cuda:0")
x = torch.randn(1, 128, 25, 25, 4, device=dev).to(torch.bfloat16).requires_grad_(True)
core = (torch.randn(*(4,) * 9, 6, device=dev) * 4.0 ** -4.5).to(torch.bfloat16).requires_grad_(True)
dy = torch.randn(128, 23, 23, 6, device=dev).to(torch.bfloat16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    x.grad = core.grad = None
    eps(core, x).backward(dy)
torch.cuda.synchronize()

import sys, os
sys.path.insert(0, "/root/repo")
import torch
from dctn_amd.eps import eps
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(1, 64, 28, 28, 2, device=dev, dtype=torch.float64, requires_grad=True)
core = torch.randn(*(2,) * 16, 2, device=dev, dtype=torch.float64, requires_grad=True)
dy = torch.randn(64, 25, 25, 2, device=dev, dtype=torch.float64)
for _ in range(10):
    x.grad = None; core.grad = None
    eps(core, x).backward(dy)
torch.cuda.synchronize()

#!/usr/bin/env python3
"""The complete cfg2 bf16 training iteration with the fused tail, replayed from its HIP graph (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd
from dctn_amd.training import FlatSGD, GraphedTrainStep, fused_cross_entropy

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, dev, torch.bfloat16)
u = torch.rand(1, 1024, 28, 28)
x = torch.stack((torch.sin(u * torch.pi / 2) ** 2, torch.cos(u * torch.pi / 2) ** 2), dim=-1).to(torch.bfloat16).to(dev)
y = torch.randint(0, 10, (1024,), device=dev)
opt = FlatSGD(list(model.epses) + [model.linear.weight], [model.linear.bias], lr=1e-3, momentum=0.9, l2=1e-2)
step = GraphedTrainStep(model, x, y, fused_cross_entropy, opt, warmup=2)
for _ in range(200):
    step(x, y)
torch.cuda.synchronize()

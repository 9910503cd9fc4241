import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch
from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSSpecCore
from dctn_amd.pos2d import Pos2D
A = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
def string(pos, mid): return tuple(SBSSpecCore(Pos2D(*p), mid if i == 4 else 1) for i, p in enumerate(pos))
dev = torch.device("cuda:0")
many = ManyConvSBS(2, 2, 4, False, (string(A, 2), string(A[::-1], 2)), (DumbNormalInitialization(0.35),) * 2).to(dev)
x = torch.rand(2, 128, 26, 26, 2, device=dev, requires_grad=True)
def fb():
    x.grad = None
    ys = many(x)
    sum(y.sum() for y in ys).backward()
for _ in range(5): fb()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): fb()
torch.cuda.synchronize()
print("fwd+bwd per call", (time.perf_counter() - t0) / 50 * 1e3, "ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(50): fb()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

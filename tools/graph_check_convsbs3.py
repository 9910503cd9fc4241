import os, sys, copy
sys.path.insert(0, os.getcwd())
import torch
from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSSpecCore
from dctn_amd.pos2d import Pos2D
from dctn_amd.training import GraphedTrainStep, train_step
A = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
Bs = [(0, 0), (1, 0), (2, 0), (2, 1), (1, 1), (0, 1), (0, 2), (1, 2), (2, 2)]
def string(pos, mid): return tuple(SBSSpecCore(Pos2D(*p), mid if i == 4 else 1) for i, p in enumerate(pos))
dev = torch.device("cuda:0")
bond = 4
class One(torch.nn.Module):
    def __init__(self, ring, specs, C):
        super().__init__()
        self.layer = ManyConvSBS(C, 2, bond, ring, specs, (DumbNormalInitialization(0.6 if C == 1 else 0.3),) * len(specs))
    def forward(self, x):
        outs = self.layer(tuple(x[i] for i in range(x.shape[0])))
        o = torch.cat([t.reshape(t.shape[0], -1, t.shape[-1]).mean(1) for t in outs], 1)   # (B, sum o)
        return torch.tanh(o * 50.0)
ce = torch.nn.functional.cross_entropy
for ring in (False, True):
    for name, specs, C in (("two strings C=1", (string(A, 2), string(Bs, 2)), 1), ("one string C=1", (string(A, 2),), 1), ("two strings C=2", (string(A, 2), string(Bs, 2)), 2), ("final C=2", (string(A, 10),), 2)):
        torch.manual_seed(5)
        a = One(ring, specs, C).to(dev)
        b = copy.deepcopy(a)
        xs = [torch.rand(C, 8, 8, 8, 2, device=dev) for _ in range(3)]
        nout = sum(s[4].out_quantum_dim_size if hasattr(s[4], "out_quantum_dim_size") else 0 for s in specs)
        ys = [torch.randint(0, 2, (8,), device=dev) for _ in range(3)]
        oa = torch.optim.SGD(a.parameters(), lr=0.05); ob = torch.optim.SGD(b.parameters(), lr=0.05)
        g = GraphedTrainStep(b, xs[0], ys[0], ce, ob, warmup=3)
        for _ in range(3): train_step(a, xs[0], ys[0], ce, oa)
        diffs = []
        for x, y in zip(xs, ys):
            ra = train_step(a, x, y, ce, oa); rb = g(x, y)
            diffs.append(abs(float(ra["loss"]) - float(rb["loss"])))
        pd = max(float((pa - pb).abs().max()) for pa, pb in zip(a.parameters(), b.parameters()))
        print("ring", ring, name, "loss diffs", ["%.2e" % d for d in diffs], "param diff %.2e" % pd)

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
class Two(torch.nn.Module):
    def __init__(self, ring, n1, n2):
        super().__init__()
        s1 = (string(A, 2), string(Bs, 2))[:n1]
        s2 = (string(A, 2), string(Bs, 2))[:n2]
        self.l1 = ManyConvSBS(1, 2, bond, ring, s1, (DumbNormalInitialization(0.6),) * n1)
        self.l2 = ManyConvSBS(2 if n1 == 2 else 1, 2, bond, ring, s2, (DumbNormalInitialization(0.3 if n1 == 2 else 0.6),) * n2)
    def forward(self, x):
        h = tuple(torch.tanh(o * 30.0) for o in self.l1((x[0],)))
        outs = self.l2(h)
        o = torch.cat([t.reshape(t.shape[0], -1, t.shape[-1]).mean(1) for t in outs], 1)
        return torch.tanh(o * 50.0)
ce = torch.nn.functional.cross_entropy
for ring in (False, True):
    for n1, n2 in ((1, 1), (2, 1), (2, 2)):
        torch.manual_seed(5)
        a = Two(ring, n1, n2).to(dev)
        b = copy.deepcopy(a)
        xs = [torch.rand(1, 8, 8, 8, 2, device=dev) for _ in range(3)]
        ys = [torch.randint(0, 2, (8,), device=dev) for _ in range(3)]
        oa = torch.optim.SGD(a.parameters(), lr=0.05); ob = torch.optim.SGD(b.parameters(), lr=0.05)
        g = GraphedTrainStep(b, xs[0], ys[0], ce, ob, warmup=3)
        for _ in range(3): train_step(a, xs[0], ys[0], ce, oa)
        diffs = []
        for x, y in zip(xs, ys):
            ra = train_step(a, x, y, ce, oa); rb = g(x, y)
            diffs.append(abs(float(ra["loss"]) - float(rb["loss"])))
        pd = [float((pa - pb).abs().max()) for pa, pb in zip(a.parameters(), b.parameters())]
        print("ring", ring, "strings", n1, n2, "loss diffs", ["%.2e" % d for d in diffs], "max param diff %.2e" % max(pd), "first differing param", next((i for i, d in enumerate(pd) if d > 0), None), "of", len(pd))

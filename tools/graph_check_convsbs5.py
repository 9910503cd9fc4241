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
ring = True
LR = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
class Classifier(torch.nn.Module):
    def __init__(self, final_out):
        super().__init__()
        init = DumbNormalInitialization((2 * bond) ** -0.5 * 1.3)
        two = (string(A, 2), string(Bs, 2))
        self.layers = torch.nn.ModuleList([ManyConvSBS(1, 2, bond, ring, two, (init,) * 2), ManyConvSBS(2, 2, bond, ring, two, (init,) * 2),
                                           ManyConvSBS(2, 2, bond, ring, (string(A, final_out),), (init,))])
        self.scales = [1.0, 1.0, 1.0]
    def forward(self, x):
        inter = (x[0],)
        for layer, scale in zip(self.layers, self.scales):
            inter = tuple(torch.tanh(o * scale) for o in layer(inter))
        (out,) = inter
        return out.reshape(out.shape[0], -1, out.shape[-1]).mean(1)
    def calibrate(self, x):
        with torch.no_grad():
            inter = (x[0],)
            for k, layer in enumerate(self.layers):
                outs = layer(inter)
                self.scales[k] = 1.0 / float(torch.cat([o.reshape(-1) for o in outs]).abs().median())
                inter = tuple(torch.tanh(o * self.scales[k]) for o in outs)
ce = torch.nn.functional.cross_entropy
for final_out, hw in ((10, 8),):
    torch.manual_seed(5)
    a = Classifier(final_out).to(dev)
    xs = [torch.rand(1, 8, hw, hw, 2, device=dev) for _ in range(4)]
    ys = [torch.randint(0, min(final_out, 10), (8,), device=dev) for _ in range(4)]
    a.calibrate(xs[0])
    b = copy.deepcopy(a)
    if len(sys.argv) > 2: y0 = a(xs[0])
    oa = torch.optim.SGD(a.parameters(), lr=LR); ob = torch.optim.SGD(b.parameters(), lr=LR)
    g = GraphedTrainStep(b, xs[0], ys[0], ce, ob, warmup=3)
    pd0 = [float((pa - pb).abs().max()) for pa, pb in zip(a.parameters(), b.parameters())]
    for _ in range(3): train_step(a, xs[0], ys[0], ce, oa)
    pd1 = [float((pa - pb).abs().max()) for pa, pb in zip(a.parameters(), b.parameters())]
    diffs = []
    for it, (x, y) in enumerate(zip(xs, ys)):
        ra = train_step(a, x, y, ce, oa); rb = g(x, y)
        if it == 0:
            for (n_, pa), pb in zip(a.named_parameters(), b.parameters()):
                ga, gb = float(pa.grad.abs().max()), float(pb.grad.abs().max())
                if not (abs(ga - gb) <= 1e-3 * max(ga, 1e-30)): print('         ', n_, tuple(pa.shape), 'eager %.3e graph %.3e' % (ga, gb))
        diffs.append(abs(float(ra["loss"]) - float(rb["loss"])))
        print("      eager %.6f graphed %.6f" % (float(ra["loss"]), float(rb["loss"])), "graph param nonfinite", sum(int(not torch.isfinite(p).all()) for p in b.parameters()), "eager param nonfinite", sum(int(not torch.isfinite(p).all()) for p in a.parameters()), "graph grad absmax %.3e" % max(float(p.grad.abs().max()) for p in b.parameters()), "eager grad absmax %.3e" % max(float(p.grad.abs().max()) for p in a.parameters()))
    pd = [float((pa - pb).abs().max()) for pa, pb in zip(a.parameters(), b.parameters())]
    print("final_out", final_out, "hw", hw, "after warmups: max param diff %.2e" % max(pd1), "first", next((i for i, d in enumerate(pd1) if d > 0), None),
          "| loss diffs", ["%.2e" % d for d in diffs], "| end param diff %.2e" % max(pd))

import os, sys, copy
sys.path.insert(0, os.getcwd())
import torch
from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSSpecCore
from dctn_amd.pos2d import Pos2D
A = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
Bs = [(0, 0), (1, 0), (2, 0), (2, 1), (1, 1), (0, 1), (0, 2), (1, 2), (2, 2)]
def string(pos, mid): return tuple(SBSSpecCore(Pos2D(*p), mid if i == 4 else 1) for i, p in enumerate(pos))
dev = torch.device("cuda:0")
bond, ring = 4, True
class Classifier(torch.nn.Module):
    def __init__(self, nlayers):
        super().__init__()
        init = DumbNormalInitialization((2 * bond) ** -0.5 * 1.3)
        two = (string(A, 2), string(Bs, 2))
        ls = [ManyConvSBS(1, 2, bond, ring, two, (init,) * 2)]
        if nlayers > 1: ls.append(ManyConvSBS(2, 2, bond, ring, two, (init,) * 2))
        if nlayers > 2: ls.append(ManyConvSBS(2, 2, bond, ring, (string(A, 10),), (init,)))
        self.layers = torch.nn.ModuleList(ls)
    def forward(self, x):
        inter = (x[0],)
        for layer in self.layers:
            inter = layer(inter)
        return sum(o.reshape(o.shape[0], -1).sum(1) for o in inter)
for nl in (1, 2, 3):
    torch.manual_seed(5)
    m = Classifier(nl).to(dev)
    x = torch.rand(1, 8, 8, 8, 2, device=dev)
    def fb():
        for c in m.parameters(): c.grad = None
        y = m(x)
        y.sum().backward()
        return y
    y_e = fb().detach().clone(); gc_e = [p.grad.clone() for p in m.parameters()]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fb()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y_g = fb()
    g.replay(); torch.cuda.synchronize()
    bad = [i for i, (p, ge) in enumerate(zip(m.parameters(), gc_e)) if not torch.allclose(p.grad, ge, rtol=1e-4, atol=1e-7 * float(ge.abs().max()))]
    print("layers", nl, "y ok", bool(torch.allclose(y_g, y_e)), "y finite", bool(torch.isfinite(y_g).all()), "bad grads", bad, "nan grads", [i for i, p in enumerate(m.parameters()) if not torch.isfinite(p.grad).all()])

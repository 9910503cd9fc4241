import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSSpecCore
from dctn_amd.pos2d import Pos2D
A = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
def string(pos, mid): return tuple(SBSSpecCore(Pos2D(*p), mid if i == 4 else 1) for i, p in enumerate(pos))
dev = torch.device("cuda:0")
torch.manual_seed(0)
for ring, C, mid in ((True, 2, 2), (False, 2, 2), (True, 2, 10)):
    many = ManyConvSBS(C, 2, 4, ring, (string(A, mid),), (DumbNormalInitialization(0.4),)).to(dev)
    x = torch.rand(C, 8, 6, 6, 2, device=dev, requires_grad=True)
    def fb():
        x.grad = None
        for c in many.parameters(): c.grad = None
        (y,) = many(x)
        y.sum().backward()
        return y
    y_e = fb().detach().clone(); gx_e = x.grad.clone(); gc_e = [p.grad.clone() for p in many.parameters()]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fb()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y_g = fb()
    for rep in range(3):
        if rep == 1:   # dirty the gradient buffers between replays: a correct graph zero-fills them itself
            for p in many.parameters(): p.grad.fill_(7.0)
            x.grad.fill_(7.0)
        g.replay(); torch.cuda.synchronize()
        print("ring", ring, "C", C, "mid", mid, "replay", rep, "y", bool(torch.allclose(y_g, y_e)), "dx", bool(torch.allclose(x.grad, gx_e)),
              "dcores", [bool(torch.allclose(p.grad, ge)) for p, ge in zip(many.parameters(), gc_e)])

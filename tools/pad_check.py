import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dctn_amd.conv_sbs import ConvSBS, DumbNormalInitialization
from dctn_amd.conv_sbs_spec import SBSSpecCore, SBSSpecString
from dctn_amd.pos2d import Pos2D
from oracle import ref_cpu as R
import dctn_amd
dev = torch.device("cuda:0")
snake = ((0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2))
for r in (6, 3):
    for ring in (False, True):
        for q in (2, 3):
            for mid in (1, 2):
                outs = [1] * 9; outs[4] = mid
                bonds = ((r if ring else 1),) + (r,) * 8
                torch.manual_seed(1)
                spec = SBSSpecString(tuple(SBSSpecCore(Pos2D(h, w), o) for (h, w), o in zip(snake, outs)), bonds, 1, q)
                m = ConvSBS(spec, DumbNormalInitialization((q * r) ** -0.5)).to(dev)
                x = torch.randn(1, 3, 8, 9, q, device=dev, requires_grad=True)
                y = m(x); dy = torch.randn_like(y); y.backward(dy)
                cores64 = [c.detach().cpu().double() for c in m.cores]
                want = R.convsbs_forward(cores64, list(snake), x.detach().cpu().double())
                gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, list(snake), xx), [x.detach().cpu().double()] + cores64, dy.cpu().double())
                def rel(a, b): return float((a.cpu().double() - b).abs().max()) / max(float(b.abs().max()), 1e-300)
                print("r", r, "ring", ring, "q", q, "mid", mid, dctn_amd.last_kernel(), "y %.1e dx %.1e" % (rel(y, want), rel(x.grad, gr[0])), "dcores", ["%.0e" % rel(c.grad, g) for c, g in zip(m.cores, gr[1:])])

import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSSpecCore
from dctn_amd.pos2d import Pos2D
from oracle import ref_cpu as R
A = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
def string(pos, mid): return tuple(SBSSpecCore(Pos2D(*p), mid if i == 4 else 1) for i, p in enumerate(pos))
dev = torch.device("cuda:0")
torch.manual_seed(0)
keep = []
def poison():
    junk = [torch.full((n,), float("nan"), device=dev) for n in (64, 512, 4096, 1 << 15, 1 << 17, 1 << 20, 1 << 22)] * 3
    keep.append(junk[::5])   # keep some alive so that the free list is fragmented
    del junk
for ring in (True, False):
    for C, mid, B, HW in ((2, 10, 8, 4), (2, 2, 8, 6), (1, 2, 8, 8), (2, 10, 8, 5), (2, 10, 3, 9)):
        many = ManyConvSBS(C, 2, 4, ring, (string(A, mid),), (DumbNormalInitialization(0.4),)).to(dev)
        sbs = many.strings[0]
        cores64 = [c.detach().cpu().double() for c in sbs.cores]
        pos = [(c.position.h, c.position.w) for c in sbs.spec.cores]
        worst = 0.0
        for trial in range(6):
            poison()
            x = torch.rand(C, B, HW, HW, 2, device=dev, requires_grad=True)
            poison()
            (y,) = many(x)
            poison()
            dy = torch.randn_like(y)
            for c in many.parameters(): c.grad = None
            y.backward(dy)
            want = R.convsbs_forward(cores64, pos, x.detach().cpu().double())
            gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, pos, xx), [x.detach().cpu().double()] + cores64, dy.cpu().double())
            def rel(a, b): return float((a.cpu().double() - b).abs().max()) / max(float(b.abs().max()), 1e-300) if torch.isfinite(a).all() else float("inf")
            errs = [rel(y, want), rel(x.grad, gr[0])] + [rel(c.grad, g) for c, g in zip(sbs.cores, gr[1:])]
            worst = max(worst, max(errs))
            if max(errs) > 1e-3: print("   trial", trial, "errs", ["%.1e" % e for e in errs])
        print("ring", ring, "C", C, "mid", mid, "B", B, "HW", HW, "windows", B * (HW - 2) ** 2, "worst rel err %.2e" % worst)

import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSSpecCore
from dctn_amd.pos2d import Pos2D
import dctn_amd
A = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
Bs = [(0, 0), (1, 0), (2, 0), (2, 1), (1, 1), (0, 1), (0, 2), (1, 2), (2, 2)]
def string(pos, mid): return tuple(SBSSpecCore(Pos2D(*p), mid if i == 4 else 1) for i, p in enumerate(pos))
dev = torch.device("cuda:0")
torch.manual_seed(5)
bond = 4
init = DumbNormalInitialization((2 * bond) ** -0.5 * 1.3)
two = (string(A, 2), string(Bs, 2))
layers = torch.nn.ModuleList([ManyConvSBS(1, 2, bond, False, two, (init,) * 2), ManyConvSBS(2, 2, bond, False, two, (init,) * 2),
                              ManyConvSBS(2, 2, bond, False, (string(A, 10),), (init,))]).to(dev)
x = torch.rand(1, 8, 8, 8, 2, device=dev)
with torch.no_grad():
    inter = (x[0],)
    for k, layer in enumerate(layers):
        outs = layer(inter)
        allv = torch.cat([o.reshape(-1) for o in outs])
        print(k, [tuple(o.shape) for o in outs], "absmax", float(allv.abs().max()), "median|o|", float(allv.abs().median()), "std", float(allv.std()), "nan", int(torch.isnan(allv).sum()), dctn_amd.last_kernel())
        # oracle check of this layer
        from oracle import ref_cpu as R
        xin = torch.stack([t.cpu().double() for t in inter]) if len(inter) > 1 else inter[0].cpu().double()[None]
        for sidx, (sbs, o) in enumerate(zip(layer.strings, outs)):
            want = R.convsbs_forward([c.detach().cpu().double() for c in sbs.cores], [(c.position.h, c.position.w) for c in sbs.spec.cores], xin)
            err = float((o.cpu().double() - want).abs().max()) / max(float(want.abs().max()), 1e-300)
            print("    string", sidx, "rel err vs oracle", err)
        sc = 1.0 / float(allv.abs().median())
        inter = tuple(o * sc for o in outs)

#!/usr/bin/env python3
"""Phase stamps of the register-resident ConvSBS backward (diagnostic build: `make -C dctn_amd/csrc EXTRA=-DDCTN_STAMPS`
after touching convsbs_reg.hip): per workgroup the s_memtime value at 0 kernel entry, 1 accumulators zeroed + barrier,
2 features loaded and forward chain done, 3 adjoint sweep done, 4 workgroup barrier passed, 5 dX written, 6 record
written.  Prints the median / max over the workgroups of every phase, in microseconds of the 100 MHz counter.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dctn_amd import _lib as L  # noqa: E402
from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS  # noqa: E402
from dctn_amd.conv_sbs_spec import SBSSpecCore  # noqa: E402
from dctn_amd.pos2d import Pos2D  # noqa: E402

SNAKE = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
dev = torch.device("cuda:0")
r, B = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 128
spec = (tuple(SBSSpecCore(Pos2D(*p), 2 if i == 4 else 1) for i, p in enumerate(SNAKE)),)
string = ManyConvSBS(1, 3, r, False, spec, (DumbNormalInitialization((3 * r) ** -0.5),)).to(dev).strings[0]
x = torch.randn(1, B, 32, 32, 3, device=dev)
sp = string.spec
n = len(sp)
cores = [c.detach().contiguous() for c in string.cores]
outs = L.int_array([s_.out_quantum_dim_size for s_ in sp.shapes])
bonds = L.int_array(sp.bond_sizes)
ph, pw = L.int_array([p.h for p in sp.positions]), L.int_array([p.w for p in sp.positions])
dy = torch.randn(B, 30, 30, 2, device=dev)
dx = torch.empty_like(x)
dcs = [torch.empty_like(c) for c in cores]
lib, code = L.lib(), L.F32
nws = lib.dctn_convsbs_workspace_bytes(n, outs, bonds, 1, B, 32, 32, 3, ph, pw, code, 1)
ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
cp, dcp, xs = L.ptr_array(cores), L.ptr_array(dcs), L.strides5(x)
for _ in range(3):
    L.check(lib.dctn_convsbs_bwd(x.data_ptr(), xs, cp, dy.data_ptr(), dx.data_ptr(), dcp, n, outs, bonds, ph, pw, 1, B, 32, 32, 3,
                                 ws.data_ptr(), ws.numel(), code, L.stream_ptr(dev)), "b")
torch.cuda.synchronize()
tot = sum(c.numel() for c in cores)
nrec = B * 2 if B * 2 >= 256 else None
for nrec in ([nrec] if nrec else range(B, 40 * B)):
    off = (tot * nrec + 63) // 64 * 64 * 4
    st = ws[off:off + nrec * 64].view(torch.int64).view(nrec, 8).cpu()
    if int(st[:, 0].min()) > 0 and int((st[:, 6] - st[:, 0]).min()) > 0:
        break
print(f"B={B} records={nrec}")
t0 = st[:, 0].min()
names = ["zero+barrier", "loads+forward", "adjoint sweep", "barrier wait", "dX phase", "record"]
for k, nm in enumerate(names):
    d = (st[:, k + 1] - st[:, k]).double() / 100.0
    print(f"  {nm:14s} median {d.median():6.2f} us   max {d.max():6.2f}   min {d.min():6.2f}")
print(f"  start skew    max {((st[:, 0] - t0).double() / 100).max():6.2f} us; kernel end (last record) {((st[:, 6].max() - t0).double() / 100):6.2f} us")

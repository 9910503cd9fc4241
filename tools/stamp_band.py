#!/usr/bin/env python3
"""Phase stamps of the band-owning ConvSBS backward (convsbs_band.hip).  Diagnostic build only, made on the GPU box:

    gpurun -- 'touch dctn_amd/csrc/convsbs_band.hip && make -C dctn_amd/csrc EXTRA=-DDCTN_STAMPS >/dev/null && python tools/stamp_band.py'

Wave 0 (chain role) and wave 4 (gradient role) of every workgroup leave the shader clock at: 0 entry, 1 packs staged,
and for the SECOND tile of the wave: 2 tile start, 3 features staged, 4 forward sweep done, 5 last core done, 6..12 middle
cores 7..1 done (each: its pair steps and the hand-over barriers), 13 first core done; 14 all tiles done, 15 barrier (A),
16 barrier (B1) (chain waves: dX written; gradient waves: first join round), 17 second join round, 18 record written.  Prints medians over the workgroups in shader cycles.
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
r = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
q = int(sys.argv[3]) if len(sys.argv) > 3 else 3
spec = (tuple(SBSSpecCore(Pos2D(*p), 2 if i == 4 else 1) for i, p in enumerate(SNAKE)),)
string = ManyConvSBS(1, q, r, False, spec, (DumbNormalInitialization((q * r) ** -0.5),)).to(dev).strings[0]
x = torch.randn(1, B, 32, 32, q, device=dev)
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
nws = lib.dctn_convsbs_workspace_bytes(n, outs, bonds, 1, B, 32, 32, q, ph, pw, code, 1)
ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
cp, dcp, xs = L.ptr_array(cores), L.ptr_array(dcs), L.strides5(x)
for _ in range(5):
    L.check(lib.dctn_convsbs_bwd(x.data_ptr(), xs, cp, dy.data_ptr(), dx.data_ptr(), dcp, n, outs, bonds, ph, pw, 1, B, 32, 32, q,
                                 ws.data_ptr(), ws.numel(), code, L.stream_ptr(dev)), "b")
torch.cuda.synchronize()
assert L.last_kernel() == "convsbs_bwd_band_f32", L.last_kernel()
tot = sum(c.numel() for c in cores)
nb = max(1, -(-256 // B))
nwg = None
host = ws.cpu()
for cand_nb in range(1, 31):   # the plan's band count is not exported: find the stamp block by its shape
    rows = -(-30 // cand_nb)
    nbb = -(-30 // rows)
    wg = B * nbb
    rec = (wg * (8 * q * 256 + 128) * 4 + 255) // 256 * 256   # a record: the join area's layout (8 slots x q tiles + first / last)
    side = (B * (nbb - 1) * 2 * 2 * 32 * q * 4 + 255) // 256 * 256
    off = rec + side
    if off + wg * 2 * 32 * 8 > host.numel():
        continue
    st = host[off:off + wg * 2 * 32 * 8].view(torch.int64).view(wg, 2, 32)
    if int(st[:, 0, 0].min()) > 0 and int((st[:, 0, 18] - st[:, 0, 0]).min()) > 0 and int((st[:, 1, 18] - st[:, 1, 0]).min()) > 0:
        nwg = wg
        break
assert nwg, "no stamps found: is this the DCTN_STAMPS build?"
print(f"bond {r} q {q} B {B}: {nwg} workgroups")
names = {1: "packs staged", 2: "(tile 2 starts)", 3: "features staged", 4: "forward sweep", 5: "last core", 13: "first core",
         14: "remaining tiles", 15: "barrier A", 16: "dX | join round 1", 17: "join round 2", 18: "record"}
for c in range(7, 0, -1):
    names[5 + (8 - c)] = f"middle core {c}"
for role, nm in ((0, "chain wave 0"), (1, "gradient wave 4")):
    s = st[:, role, :].double()
    print(f" {nm}: kernel {float((s[:, 18] - s[:, 0]).median()):9.0f} cycles (median), max {float((s[:, 18] - s[:, 0]).max()):9.0f}")
    prev = 0
    for k in range(1, 19):
        if role == 1 and k in (3, 4, 5, 13, 17):
            continue
        if float(s[:, k].min()) <= 0:
            continue
        d = s[:, k] - s[:, prev]
        print(f"   {k:2d} {names.get(k, ''):18s} median {float(d.median()):8.0f}   max {float(d.max()):8.0f}   min {float(d.min()):8.0f}")
        prev = k
    fine = {0: ["19 core 3 starts", "20 hand-over stored, A operand read, 12 MFMAs issued", "21 barrier passed", "22 epilogue (dv, df) done",
                "23 feature gradient summed over the k groups and put in its row"],
            1: ["19 arrives at the barrier of core 3's hand-over", "20 barrier passed, operands of this hand-over read", "21 pending hand-over multiplied (12 MFMAs issued)"]}[role]
    for k in range(20, 19 + len(fine)):
        d = s[:, k] - s[:, k - 1]
        print(f"      {fine[k - 19]:45s} median {float(d.median()):7.0f}   max {float(d.max()):7.0f}")
    if role == 0:
        print(f"   tile 2 in all (2 -> 13): {float((s[:, 13] - s[:, 2]).median()):8.0f}; forward share {float(((s[:, 4] - s[:, 2]) / (s[:, 13] - s[:, 2])).median()):.2f}")

#!/usr/bin/env python3
"""Device time of the ConvSBS C-ABI calls of the cfg4 string (9-core snake, CIFAR layout) with either gradient switched
off - where the backward's time goes (dCore reduction vs dX gather) - at several batch sizes (fixed cost vs work).

    python tools/time_convsbs_calls.py [bond]
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
r = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def graph_time(fn, iters=50, per=8):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / per * 1e3


for B in (16, 64, 128, 256):
    spec = (tuple(SBSSpecCore(Pos2D(*p), 2 if i == 4 else 1) for i, p in enumerate(SNAKE)),)
    torch.manual_seed(0)
    string = ManyConvSBS(1, 3, r, False, spec, (DumbNormalInitialization((3 * r) ** -0.5),)).to(dev).strings[0]
    x = torch.randn(1, B, 32, 32, 3, device=dev)
    sp = string.spec
    n = len(sp)
    cores = [c.detach().contiguous() for c in string.cores]
    outs = L.int_array([s_.out_quantum_dim_size for s_ in sp.shapes])
    bonds = L.int_array(sp.bond_sizes)
    ph, pw = L.int_array([p.h for p in sp.positions]), L.int_array([p.w for p in sp.positions])
    out = torch.empty(B, 30, 30, 2, device=dev)
    dy = torch.randn_like(out)
    dx = torch.empty_like(x)
    dcs = [torch.empty_like(c) for c in cores]
    lib, code = L.lib(), L.F32
    wsb = torch.empty(max(256, lib.dctn_convsbs_workspace_bytes(n, outs, bonds, 1, B, 32, 32, 3, ph, pw, code, 1)), dtype=torch.uint8, device=dev)
    wsf = torch.empty(max(256, lib.dctn_convsbs_workspace_bytes(n, outs, bonds, 1, B, 32, 32, 3, ph, pw, code, 0)), dtype=torch.uint8, device=dev)
    cp, dcp, xs = L.ptr_array(cores), L.ptr_array(dcs), L.strides5(x)
    st = lambda: L.stream_ptr(dev)

    def fwd():
        L.check(lib.dctn_convsbs_fwd(x.data_ptr(), xs, cp, out.data_ptr(), n, outs, bonds, ph, pw, 1, B, 32, 32, 3, wsf.data_ptr(), wsf.numel(), code, st()), "f")

    def bwd(want_dx, want_dc):
        def go():
            L.check(lib.dctn_convsbs_bwd(x.data_ptr(), xs, cp, dy.data_ptr(), dx.data_ptr() if want_dx else None, dcp if want_dc else None,
                                         n, outs, bonds, ph, pw, 1, B, 32, 32, 3, wsb.data_ptr(), wsb.numel(), code, st()), "b")
        return go

    print(f"B={B:4d} bond {r}: fwd {graph_time(fwd):6.1f} us | bwd dX+dCore {graph_time(bwd(True, True)):6.1f} | dX only {graph_time(bwd(True, False)):6.1f} "
          f"| dCore only {graph_time(bwd(False, True)):6.1f}   [{L.last_kernel()}]", flush=True)

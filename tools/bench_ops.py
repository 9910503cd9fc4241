#!/usr/bin/env python3
"""Per-op measurements for the SURVEY 8(d) configs that are not the headline bench line:
ConvSBS (cfg4), logmatmulexp fold (cfg5), single EPS layers (cfg1/cfg3).  HIP events around
forward and forward+backward (dctn/benchmark.py protocol), optional CPU oracle timing.

    python tools/bench_ops.py [--cpu] [--only convsbs|lme|eps]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import dctn_amd  # noqa: E402

DEV = torch.device("cuda:0")


def time_gpu(fn, iters):
    """Median over `iters` rounds of 3 back-to-back calls (an occasional allocator stall of tens of
    milliseconds otherwise lands in a mean)."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    rounds = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        rounds.append(e0.elapsed_time(e1) / 3 * 1e-3)
    rounds.sort()
    return rounds[len(rounds) // 2]


def time_cpu(fn, budget=6.0):
    fn()
    t0 = time.perf_counter()
    fn()
    one = time.perf_counter() - t0
    n = max(1, min(50, int(budget / max(one, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


def report(name, windows, fwd_s, fb_s, extra=None):
    row = {"op": name, "windows": windows, "fwd_us": round(fwd_s * 1e6, 1), "fwd_bwd_us": round(fb_s * 1e6, 1),
           "fwd_Mwin_s": round(windows / fwd_s / 1e6, 1), "fwd_bwd_Mwin_s": round(windows / fb_s / 1e6, 1)}
    if extra:
        row.update(extra)
    print(json.dumps(row), flush=True)


def bench_convsbs(cpu):
    from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
    from dctn_amd.conv_sbs_spec import SBSSpecCore
    from dctn_amd.pos2d import Pos2D
    from oracle import ref_cpu as R

    snake = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
    spec = (tuple(SBSSpecCore(Pos2D(*p), 2 if i == 4 else 1) for i, p in enumerate(snake)),)
    for (C, q, HW, tag) in ((1, 3, 32, "cfg4 CIFAR colour q=3"), (2, 2, 30, "cfg4 second layer C=2 q=2")):
        for r in (4, 8, 16):
            B = 128
            torch.manual_seed(r)
            many = ManyConvSBS(C, q, r, False, spec, (DumbNormalInitialization((q**C * r) ** -0.5),)).to(DEV)
            x = torch.randn(C, B, HW, HW, q, device=DEV, requires_grad=True)
            (y,) = many(x)
            dy = torch.randn_like(y)
            windows = y.shape[0] * y.shape[1] * y.shape[2]

            def fwd():
                with torch.no_grad():
                    many(x)

            def fb():
                x.grad = None
                for prm in many.parameters():
                    prm.grad = None
                many(x)[0].backward(dy)

            extra = {}
            if cpu:
                Bc = 16
                cores = [c.detach().cpu() for c in many.strings[0].cores]
                xc = x.detach().cpu()[:, :Bc].clone().requires_grad_(True)
                cc = [c.clone().requires_grad_(True) for c in cores]
                dyc = dy.cpu()[:Bc]
                torch.set_num_threads(min(16, os.cpu_count() or 1))
                t = time_cpu(lambda: R.convsbs_forward(cc, snake, xc).backward(dyc))
                extra["cpu_fwd_bwd_Mwin_s"] = round(windows * Bc / B / t / 1e6, 3)
            tf, tb = time_gpu(fwd, 20), time_gpu(fb, 10)
            extra["bwd_kernel"] = dctn_amd.last_kernel()
            report(f"ConvSBS snake r={r} {tag} B={B} f32", windows, tf, tb, extra)


def bench_lme(cpu):
    from dctn_amd.logmatmulexp import logmatmulexp, logmatmulexp_fold
    from oracle import ref_cpu as R
    import functools

    for Wn in (86528, 692224):  # 128 and 1024 samples of 26x26 windows (cfg5)
        m = torch.randn(Wn, 9, 16, 16, device=DEV, requires_grad=True)
        y = logmatmulexp_fold(m)
        dy = torch.randn_like(y)

        def fwd():
            with torch.no_grad():
                logmatmulexp_fold(m)

        def fb():
            m.grad = None   # time the op, not a 6 GB accumulate into an old .grad
            logmatmulexp_fold(m).backward(dy)

        extra = {"fwd_GBs": None}
        f = time_gpu(fwd, 10)
        extra["fwd_GBs"] = round(Wn * 10240 / f / 1e9, 1)
        if cpu and Wn < 100000:
            mc = m.detach().cpu()[:4096].clone().requires_grad_(True)
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            t = time_cpu(lambda: R.logmatmulexp_fold_batched(mc).backward(dy.cpu()[:4096]))
            extra["cpu_fwd_bwd_Mwin_s"] = round(4096 / t / 1e6, 3)
        report(f"logmatmulexp fold 9x(16x16) f32 windows={Wn}", Wn, f, time_gpu(fb, 5), extra)
    # the reference's own benchmark: reduce(logmatmulexp, 6 x (256x256)) f32 (results.json row 1)
    mats = [torch.randn(256, 256, device=DEV, requires_grad=True) for _ in range(6)]
    y = functools.reduce(logmatmulexp, mats)
    dy = torch.randn_like(y)
    f = time_gpu(lambda: functools.reduce(logmatmulexp, [t.detach() for t in mats]), 20)
    fb = time_gpu(lambda: functools.reduce(logmatmulexp, mats).backward(dy), 10)
    print(json.dumps({"op": "reduce(logmatmulexp, 6x(256x256)) f32", "fwd_ms": round(f * 1e3, 3),
                      "fwd_bwd_ms": round(fb * 1e3, 3),
                      "reference_published_ms": {"fwd": 5.51, "fwd_bwd": 11.08, "hardware": "unnamed CUDA GPU"}}),
          flush=True)
    # the other rows the reference publishes (results.json:372-381 and :822-831)
    for dim, dt, pub in ((128, torch.float32, {"fwd": 0.815, "fwd_bwd": 1.676}), (280, torch.float64, {"fwd_bwd": 41.1})):
        mats = [torch.randn(dim, dim, device=DEV, dtype=dt, requires_grad=True) for _ in range(6)]
        y = functools.reduce(logmatmulexp, mats)
        dy = torch.randn_like(y)
        f = time_gpu(lambda: functools.reduce(logmatmulexp, [t.detach() for t in mats]), 20)
        fb = time_gpu(lambda: functools.reduce(logmatmulexp, mats).backward(dy), 10)
        print(json.dumps({"op": f"reduce(logmatmulexp, 6x({dim}x{dim})) {str(dt).split('.')[-1]}",
                          "fwd_ms": round(f * 1e3, 3), "fwd_bwd_ms": round(fb * 1e3, 3),
                          "fwd_kernel": dctn_amd.last_kernel(),
                          "reference_published_ms": dict(pub, hardware="unnamed CUDA GPU")}), flush=True)


def bench_eps(cpu):
    from dctn_amd.eps import eps
    from oracle import ref_cpu as R

    cases = [("cfg1 K=4 Q=2 O=2 f64 B=64", 1, 64, 28, 2, 4, 2, torch.float64),
             ("cfg3a-L1 K=4 Q=2 O=4 f32 B=128", 1, 128, 28, 2, 4, 4, torch.float32),
             ("cfg3a-L2 K=3 Q=4 O=6 f32 B=128", 1, 128, 25, 4, 3, 6, torch.float32),
             ("cfg3b-L1 K=4 Q=2 O=8 f32 B=128", 1, 128, 28, 2, 4, 8, torch.float32),
             ("cfg3b-L2 K=2 Q=8 O=8 f32 B=128", 1, 128, 25, 8, 2, 8, torch.float32),
             ("cfg3a-L1 K=4 Q=2 O=4 bf16 B=128 (two-halves bf16 MFMA path)", 1, 128, 28, 2, 4, 4, torch.bfloat16),
             ("cfg3a-L2 K=3 Q=4 O=6 bf16 B=128 (two-halves bf16 MFMA path)", 1, 128, 25, 4, 3, 6, torch.bfloat16),
             ("K=3 Q=4 O=8 bf16 B=128 (two-halves bf16 MFMA path, fused epilogue)", 1, 128, 25, 4, 3, 8, torch.bfloat16),
             ("odd Q: C=2 K=2 Q=3 O=8 f32 B=128 (two-halves f32 path)", 2, 128, 28, 3, 2, 8, torch.float32),
             ("odd Q: C=1 K=3 Q=3 O=6 f32 B=128 (two-halves f32 path)", 1, 128, 28, 3, 3, 6, torch.float32)]
    for name, C, B, HW, Q, K, O, dt in cases:
        N = K * K * C
        x = torch.randn(C, B, HW, HW, Q, device=DEV, dtype=dt, requires_grad=True)
        core = (torch.randn(*(Q,) * N, O, device=DEV, dtype=dt) * Q ** (-N / 2)).requires_grad_(True)
        y = eps(core, x)
        dy = torch.randn_like(y)
        windows = y.shape[0] * y.shape[1] * y.shape[2]
        flops = 2 * Q**N * O * windows

        def fwd():
            with torch.no_grad():
                eps(core, x)

        def fb():
            x.grad = None
            core.grad = None
            eps(core, x).backward(dy)

        f, b = time_gpu(fwd, 5), time_gpu(fb, 3)
        report("eps " + name, windows, f, b, {"kernel": dctn_amd.last_kernel(), "fwd_TFLOPs": round(flops / f / 1e12, 2),
                                                "fwd_bwd_TFLOPs": round(3 * flops / b / 1e12, 2)})


def bench_train():
    """A whole training iteration of BASELINE cfg2 (forward, CE loss, L2 regulariser, backward, SGD with
    momentum; B = 1024): eager launches vs the HIP-graph replay of dctn_amd.training.GraphedTrainStep."""
    import torch.nn.functional as F
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd
    from dctn_amd.training import GraphedTrainStep, train_step

    for dtype in (torch.bfloat16, torch.float32):
        torch.manual_seed(0)
        model = EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, DEV, dtype)
        u = torch.rand(1, 1024, 28, 28)
        x = torch.stack((torch.sin(u * torch.pi / 2) ** 2, torch.cos(u * torch.pi / 2) ** 2), dim=-1).to(dtype).to(DEV)
        y = torch.randint(0, 10, (1024,), device=DEV)
        reg = lambda m: m.epswise_l2_regularizer()
        opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)
        eager = time_gpu(lambda: train_step(model, x, y, F.cross_entropy, opt, reg_fn=reg, reg_coeff=1e-4), 20)
        graphed = GraphedTrainStep(model, x, y, F.cross_entropy, opt, reg_fn=reg, reg_coeff=1e-4)
        replay = time_gpu(lambda: graphed(x, y), 50)
        # fused tail: one CE forward + one CE backward kernel, regulariser + momentum update in one kernel
        from dctn_amd.training import FlatSGD, fused_cross_entropy
        torch.manual_seed(0)
        model2 = EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, DEV, dtype)
        opt2 = FlatSGD(list(model2.epses) + [model2.linear.weight], [model2.linear.bias], lr=1e-3, momentum=0.9, l2=1e-4)
        graphed2 = GraphedTrainStep(model2, x, y, fused_cross_entropy, opt2)
        fused = time_gpu(lambda: graphed2(x, y), 50)
        print(json.dumps({"op": f"training iteration cfg2 B=1024 {str(dtype).replace('torch.', '')} (fwd + CE + L2 reg + bwd + SGD)",
                          "windows": 692224, "eager_us": round(eager * 1e6, 1), "hip_graph_us": round(replay * 1e6, 1),
                          "hip_graph_fused_tail_us": round(fused * 1e6, 1),
                          "fused_tail_Gwin_s": round(692224 / fused / 1e9, 2)}), flush=True)


def bench_window_stats(cpu):
    """calc_scaling_factor's statistics at the reference's size (10 880 MNIST samples,
    float64): the one-pass HIP kernel, the reference's materialising formulation run on the GPU with
    torch (make_windows: K*K copies of the data), and the CPU oracle."""
    from dctn_amd.align import make_windows
    from dctn_amd.window_stats import apply_feature_map, window_mean_var
    from oracle import ref_cpu as R

    torch.manual_seed(0)
    x = apply_feature_map(torch.rand(10880, 28, 28, dtype=torch.float64)).to(DEV)
    for K in (3, 4):
        windows = 10880 * (29 - K) ** 2
        t_kernel = time_gpu(lambda: window_mean_var(x, K), 10)

        def materialise():
            w = make_windows(x, K)
            return w.mean_over_batch(), w.var_over_batch()

        t_torch = time_gpu(materialise, 3)
        row = {"op": f"window statistics K={K} f64, 10880x28x28x2 (calc_scaling_factor)", "windows": windows,
               "hip_kernel_us": round(t_kernel * 1e6, 1), "torch_materialising_gpu_us": round(t_torch * 1e6, 1),
               "hip_GBs": round(x.numel() * 8 / t_kernel / 1e9, 1)}
        if cpu:
            xc = x.cpu()
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            row["cpu_oracle_us"] = round(time_cpu(lambda: R.window_mean_var_factor(xc, K), 4.0) * 1e6, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if a.only in ("", "train"):
        bench_train()
    if a.only in ("", "stats"):
        bench_window_stats(a.cpu)
    if a.only in ("", "eps"):
        bench_eps(a.cpu)
    if a.only in ("", "convsbs"):
        bench_convsbs(a.cpu)
    if a.only in ("", "lme"):
        bench_lme(a.cpu)

#!/usr/bin/env python3
"""Headline benchmark: EPS-contraction windows/s, forward + backward, on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): `EPSesPlusLinear(((3,4),), p=1)` on
MNIST-shaped synthetic input x (1, B, 28, 28, 2) = (sin^2, cos^2)(pi u / 2), bf16 tensors with f32
accumulation, B = 1024 per GPU.  A step follows dctn/benchmark.py:40-43: `model(x).backward(out_grad)`
with a fixed `out_grad = randn_like(out)`; with N > 1 every rank runs its own batch (weak scaling)
and the step ends with the flat-bucket RCCL all-reduce of the parameter gradients.
A window = one output site of one sample: 26*26 = 676 windows per sample.

Prints ONE JSON line on rank 0 (see README / DESIGN.md section 6 for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# MI355X_MICROARCH.md: dense MFMA peaks (no sparsity): bf16 ~2.5 PFLOP/s; exact f32 157.3 TFLOP/s
MFMA_PEAK_TFLOPS = {"bfloat16": 2500.0, "float32": 157.3, "float64": 78.6}

WORKLOADS = {
    # name: (epses_specs, image_size, Q0, dtype)
    "cfg2": (((3, 4),), 28, 2, torch.bfloat16),
    "cfg2_f32": (((3, 4),), 28, 2, torch.float32),
    "cfg3a": (((4, 4), (3, 6)), 28, 2, torch.float32),
    "cfg3a_bf16": (((4, 4), (3, 6)), 28, 2, torch.bfloat16),   # bf16 storage, exact-f32 matrix-core arithmetic
    "cfg3b": (((4, 8), (2, 8)), 28, 2, torch.float32),
    "cfg3b_bf16": (((4, 8), (2, 8)), 28, 2, torch.bfloat16),
}


def synthetic_input(batch, image_size, q0, dtype, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    u = torch.rand(1, batch, image_size, image_size, generator=g)
    if q0 == 2:  # dataset_loading.py:33-36,63 feature map with nu = 1
        x = torch.stack((torch.sin(u * torch.pi / 2) ** 2, torch.cos(u * torch.pi / 2) ** 2), dim=-1)
    else:
        x = torch.randn(1, batch, image_size, image_size, q0, generator=g)
    return x.to(dtype).to(device)


def windows_per_sample(specs, image_size):
    total, side = 0, image_size
    for k, _ in specs:
        side = side - k + 1
        total += side * side
    return total


def usable_cores():
    """Host cores this process may really use: cgroup quota if there is one (a 1-GPU box gets a
    16-core share of a 256-thread host), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cap = int(os.environ.get("DCTN_BENCH_CPU_THREADS", "0"))
    return cap if cap > 0 else min(n, 64)


def cpu_baseline(specs, image_size, q0, target_seconds=12.0):
    """The oracle's restatement of the reference's 4-step path (oracle/ref_cpu.py), float32, all
    host cores, same fwd+bwd protocol, on a bounded sample of the workload."""
    from oracle import ref_cpu as R

    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    batch = 128
    x = synthetic_input(batch, image_size, q0, torch.float32, "cpu", 0)
    epses, in_size = [], q0
    for k, o in specs:
        epses.append((torch.randn(*(in_size,) * (k * k), o) * in_size ** (-k * k / 2)).requires_grad_(True))
        in_size = o
    side = image_size - sum(k for k, _ in specs) + len(specs)
    weight = (torch.randn(10, side * side * in_size) * 0.01).requires_grad_(True)
    bias = torch.zeros(10, requires_grad=True)

    def step():
        out = R.eps_plus_linear_forward(epses, weight, bias, x)
        out.backward(out_grad)

    out_grad = torch.randn(batch, 10)
    step()  # warm-up
    t0 = time.perf_counter()
    step()
    one = time.perf_counter() - t0
    iters = max(1, min(200, int(target_seconds / max(one, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = (time.perf_counter() - t0) / iters
    wps = windows_per_sample(specs, image_size) * batch / dt
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": wps, "unit": "windows/s", "cores": cores, "kind": "port",
        "sample": f"oracle 4-step path (torch CPU f32, {cores} threads, {model}), batch {batch}, "
                  f"{iters} fwd+bwd iterations, {dt*1e3:.1f} ms/iteration",
    }


def kernel_roofline(model, x, specs, image_size, steps):
    """Times the EPS kernels of the first layer alone with HIP events on the stream they are
    launched on (torch's current stream) and prices the dominant one against the compute (MFMA)
    roofline SURVEY 8(d) assigns to this path; the HBM figures ride along."""
    from dctn_amd import _lib as L

    dev = x.device
    core = model.epses[0].detach().contiguous()
    from dctn_amd.eps import _bf16_through_f32
    if _bf16_through_f32(core, x):   # bf16 storage, exact-f32 matrix-core arithmetic: time what eps() runs
        core, x = core.float(), x.float()
    C, B, H, W, Q = x.shape
    K, O = specs[0]
    Ho = H - K + 1
    out = torch.empty((B, Ho, Ho, O), dtype=x.dtype, device=dev)
    dy = torch.randn((B, Ho, Ho, O), device=dev).to(x.dtype)
    dcore = torch.empty_like(core)
    code, prec = L.dtype_code(x), L.precision()
    ws = L.workspace(L.lib().dctn_eps_bwd_workspace_bytes(C, B, H, W, Q, K, O, code, prec, 0, 1), dev)
    lib, st = L.lib(), L.stream_ptr(dev)
    esz = x.element_size()

    wsf = L.workspace(L.lib().dctn_eps_fwd_workspace_bytes(C, B, H, W, Q, K, O, code, prec), dev)

    def fwd():
        L.check(lib.dctn_eps_fwd(x.data_ptr(), L.strides5(x), core.data_ptr(), out.data_ptr(), wsf.data_ptr(),
                                 wsf.numel(), C, B, H, W, Q, K, O, code, prec, L.stream_ptr(dev)), "fwd")

    # The model's own backward for this shape: when the layer feeds the linear head directly and is in
    # the fused family (bf16 cfg2), that is dctn_eps_head_bwd (dCore + dWeight + dBias, dY formed on the
    # fly); otherwise the plain dctn_eps_bwd.
    from dctn_amd.eps_plus_linear import _EpsLinearHeadFunction

    w_head, b_head = model.linear.weight.detach().contiguous(), model.linear.bias.detach().contiguous()
    fused = len(specs) == 1 and _EpsLinearHeadFunction.supported(core, x, w_head, b_head)
    cout = w_head.shape[0]
    if fused:
        feat = out.view(B, -1)
        dl = (torch.randn((B, cout), device=dev) * 0.1).to(x.dtype)
        dw, db = torch.empty_like(w_head), torch.empty_like(b_head)
        wsh = L.workspace(lib.dctn_eps_head_bwd_workspace_bytes(C, B, H, W, Q, K, O, cout, code, prec), dev)

        def bwd():
            L.check(lib.dctn_eps_head_bwd(x.data_ptr(), L.strides5(x), feat.data_ptr(), dl.data_ptr(), w_head.data_ptr(),
                                          dcore.data_ptr(), dw.data_ptr(), db.data_ptr(), wsh.data_ptr(), wsh.numel(),
                                          C, B, H, W, Q, K, O, cout, code, prec, L.stream_ptr(dev)), "head bwd")
    else:
        def bwd():
            L.check(lib.dctn_eps_bwd(x.data_ptr(), L.strides5(x), core.data_ptr(), dy.data_ptr(), None, dcore.data_ptr(),
                                     ws.data_ptr(), ws.numel(), C, B, H, W, Q, K, O, code, prec, L.stream_ptr(dev)), "bwd")

    # (a) whole C-ABI call, back to back on torch's current stream (the stream the kernels are
    # launched on): a THROUGHPUT figure, successive launches may overlap head and tail;
    # (b) the call's dominant KERNEL alone (dctn_profile_main_kernel_only) as a chain of CHAIN dependent
    # launches captured in a HIP graph and replayed: each launch waits for the previous one to drain, as
    # it does inside the real step, which is the per-launch duration rocprofv3 reports (plus the
    # ~1 us boundary between two graph nodes).
    def timed(fn, n):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n * 1e-3

    CHAIN = 20

    def timed_chain(fn, n):
        fn()
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            for _ in range(CHAIN):
                fn()
        reps = max(1, n // CHAIN)
        return timed(graph.replay, reps) / CHAIN

    res, single = {}, {}
    kernel_symbol = {"eps_fwd": "eps_fwd_q2reg_k", "eps_bwd_dcore": "eps_bwd_dcore_q2reg_k"}
    n = max(steps, 100)
    for name, fn in (("eps_fwd", fwd), ("eps_bwd_dcore", bwd)):
        res[name] = (timed(fn, n), L.last_kernel())
        if "q2reg" in L.last_kernel():
            lib.dctn_profile_main_kernel_only(1)
            try:
                try:
                    single[name] = timed_chain(fn, n)
                except Exception as e:   # capture refused (e.g. another thread's HIP call): throughput figure
                    print(f"[bench] chain capture failed ({type(e).__name__}: {e}); back-to-back timing", file=sys.stderr)
                    torch.cuda.synchronize(dev)
                    single[name] = timed(fn, n)
            finally:
                lib.dctn_profile_main_kernel_only(0)
    wn = B * Ho * Ho
    n_in = core.numel() // O
    # this path is compute bound (MFMA for the core GEMM, VALU for the Khatri-Rao halves);
    # algorithmic flops per window: forward 2*Q^N*O (GEMM) + the two halves and the final dot; dCore the
    # same GEMM size transposed (+ forming dY and dWeight when the head is fused: 4*Cout*O)
    half = 2 * (Q ** ((K * K * C + 1) // 2) + Q ** ((K * K * C) // 2)) + 2 * Q ** ((K * K * C) // 2) * O
    flops = {"eps_fwd": wn * (2 * n_in * O + half),
             "eps_bwd_dcore": wn * (2 * n_in * O + half + (4 * cout * O if fused else 0))}
    # algorithmic bytes per launch: read x once, write out (fwd) / read dY or the features (bwd) once,
    # core / dCore once (+ dLogits, head weight and its gradient when fused)
    bytes_x = C * B * H * W * Q * esz
    bytes_y = wn * O * esz
    bytes_core = core.numel() * esz
    alg = {"eps_fwd": bytes_x + bytes_y + bytes_core,
           "eps_bwd_dcore": bytes_x + bytes_y + bytes_core + ((B * cout + 2 * w_head.numel() + cout) * esz if fused else 0)}
    dom = max(single, key=lambda k: single[k]) if single else max(res, key=lambda k: res[k][0])
    sec = single.get(dom, res[dom][0])
    kname = kernel_symbol[dom] if dom in single else res[dom][1]
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"{kname}:B{B}")
        except Exception:
            traffic = None
    peak = MFMA_PEAK_TFLOPS[str(x.dtype).replace("torch.", "")]
    achieved = flops[dom] / sec / 1e12
    return {
        "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
        "traffic": traffic, "kernel": kname, "launch_us": sec * 1e6, "algorithmic_flops": flops[dom],
        "flops_per_window": flops[dom] / wn, "algorithmic_bytes": alg[dom], "bytes_per_window": alg[dom] / wn,
        "hbm_gbs": alg[dom] / sec / 1e9, "hbm_frac": alg[dom] / sec / 1e9 / HBM_PEAK_GBS, "fused_head": bool(fused),
        "calls_us": {k: v[0] * 1e6 for k, v in res.items()},
        "kernels_us": {kernel_symbol[k]: v * 1e6 for k, v in single.items()},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (default 1024 for cfg2, 128 for cfg3)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --batch samples per GPU whatever N; strong: --batch is the GLOBAL batch, split over the N GPUs")
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a captured HIP graph")
    ap.add_argument("--graph-allreduce", type=int, default=0,
                    help="1: capture the gradient all-reduce into the step's HIP graph as well (opt-in: a capture that "
                         "fails cannot be recovered from inside the process, see the comment at try_capture)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); 'gloo' + "
                    "DCTN_BENCH_ONE_DEVICE=1 rehearses the multi-rank path on a single GPU")
    args = ap.parse_args()

    from dctn_amd import ddp
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd
    import dctn_amd

    one_device = os.environ.get("DCTN_BENCH_ONE_DEVICE") == "1"
    if one_device:  # rehearsal only: every rank on cuda:0, collectives through gloo
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
    # DCTN_BENCH_FORCE_ALLREDUCE=1: create the process group and issue the gradient all-reduce even with
    # one rank (rehearsal of the RCCL calls on a one-GPU machine; not a measurement)
    force_reduce = os.environ.get("DCTN_BENCH_FORCE_ALLREDUCE") == "1"
    # RCCL prints a version banner on stdout when its communicator comes up; stdout must carry the one JSON line
    # only, so file descriptor 1 points at stderr until the first collective has run
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        rank, local_rank, world = ddp.init_from_env(args.backend, single_rank_group=force_reduce)
        if dist.is_initialized():
            dev0 = torch.device("cuda", 0 if os.environ.get("DCTN_BENCH_ONE_DEVICE") == "1" else local_rank)
            probe = torch.zeros(1, device=dev0)
            dist.all_reduce(probe)
            torch.cuda.synchronize(dev0)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", 0 if one_device else local_rank)
    torch.cuda.set_device(dev)
    specs, image_size, q0, dtype = WORKLOADS[args.workload]
    batch = args.batch or (1024 if args.workload.startswith("cfg2") else 128)
    if args.scaling == "strong":
        batch = max(1, batch // world)

    torch.manual_seed(0)
    model = EPSesPlusLinear(specs, UnitTheoreticalOutputStd(), 1.0, dev, dtype, image_size=image_size, Q_0=q0)
    ddp.broadcast_parameters(model.parameters())
    x = synthetic_input(batch, image_size, q0, dtype, dev, seed=1 + rank)  # resident in HBM before timing
    out_grad = torch.randn(batch, 10, device=dev).to(dtype)
    reducer = (ddp.FlatGradAllReducer(model.parameters(), skip_single_rank=not force_reduce)
               if (world > 1 or force_reduce) else None)

    def fwd_bwd():
        for p in model.parameters():
            p.grad = None
        model(x).backward(out_grad)

    # The step replays from a HIP graph of fwd + bwd; the gradient all-reduce follows it eagerly on the same stream.
    # --graph-allreduce 1 captures the collective into the same graph (one launch per step from the host; measured
    # on one rank over RCCL: 51.7 -> 42.3 us/step) and checks the captured step against the eager one before trusting
    # it.  It is opt-in: when a capture fails (reproduced with gloo, whose collectives cannot be captured) the HIP
    # runtime stays in a state in which later collectives return "invalid argument" even after ending the capture
    # and switching streams, so there is no safe in-process fallback for a run that must produce a number.
    def try_capture(with_reduce):
        def body():
            fwd_bwd()
            if with_reduce:
                reducer()
        prev = torch.cuda.current_stream(dev)
        g = None
        try:
            side = torch.cuda.Stream(dev)
            side.wait_stream(prev)
            with torch.cuda.stream(side):
                for _ in range(3):
                    body()
            prev.wait_stream(side)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            # a stream of its own for every attempt: torch.cuda.graph's default capture stream is shared, and a failed
            # capture leaves it invalidated
            with torch.cuda.graph(g, stream=torch.cuda.Stream(dev), capture_error_mode="thread_local"):
                body()
            return g
        except Exception as e:  # keep measuring, and say so in the JSON line
            print(f"[bench] HIP graph capture (all-reduce inside: {with_reduce}) failed ({type(e).__name__}: {e})",
                  file=sys.stderr)
            # A capture that dies inside the `with` leaves the thread on the capture stream, still capturing
            # (torch.cuda.graph.__exit__ stops at the failing capture_end): end it and give the thread its stream back,
            # or every later call fails with "operation not permitted when stream is capturing".
            if g is not None:
                try:
                    g.capture_end()
                except Exception:
                    pass
            torch.cuda.set_stream(prev)
            try:
                torch.cuda.synchronize(dev)
            except Exception:
                pass
            return None

    graph, reduce_in_graph = None, False
    if args.graph:
        want_reduce = reducer is not None and args.graph_allreduce == 1
        if want_reduce:
            fwd_bwd()
            reducer()
            torch.cuda.synchronize(dev)
            want = [p.grad.detach().float().clone() for p in model.parameters()]
            graph = try_capture(True)

            def all_ranks(flag: bool) -> bool:   # every rank must take the same branch: the collectives must pair up
                t = torch.tensor([1.0 if flag else 0.0], device=dev)
                if world > 1:
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return bool(t.item() > 0.5)

            if not all_ranks(graph is not None):
                graph = None
            if graph is not None:
                graph.replay()
                torch.cuda.synchronize(dev)
                same = all(torch.allclose(p.grad.float(), w, rtol=2e-2, atol=1e-6 + 2e-2 * float(w.abs().max()))
                           for p, w in zip(model.parameters(), want))
                if all_ranks(same):
                    reduce_in_graph = True
                else:
                    print("[bench] captured step with all-reduce disagrees with the eager step; not using it", file=sys.stderr)
                    graph = None
        if graph is None:
            graph = try_capture(False)

    def step():
        if graph is not None:
            graph.replay()
        else:
            fwd_bwd()
        if reducer is not None and not reduce_in_graph:
            reducer()

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    kernel_used = dctn_amd.last_kernel()
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    windows_step = windows_per_sample(specs, image_size) * batch * world
    line = {
        "metric": "EPS-contraction windows/sec (fwd+bwd)",
        "value": windows_step * args.steps / elapsed,
        "unit": "windows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": {torch.bfloat16: "bf16", torch.float32: "f32"}[dtype],
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: EPSesPlusLinear({specs}) on MNIST-shaped {image_size}x{image_size} Q0={q0}, "
                        f"fwd + bwd(out_grad), batch {batch}/GPU",
            "windows_per_step": windows_step,
            "per_gpu_batch": batch,
            "parallelism": f"dp{world}",
            "hip_graph": graph is not None,
            "allreduce_in_graph": reduce_in_graph,
            "last_kernel": kernel_used,
            "grad_allreduce": None if reducer is None else (
                "in place on the backward's flat gradient buffer (1 launch)" if getattr(reducer, "_flat_key", None)
                else "gather -> all_reduce -> scatter (3 launches)"),
        },
    }
    if rank == 0:
        line["roofline"] = kernel_roofline(model, x, specs, image_size, args.steps)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(specs, image_size, q0)
        print(json.dumps(line), flush=True)
    barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

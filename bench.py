#!/usr/bin/env python3
"""Headline benchmark: EPS-contraction windows/s, forward + backward, on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json configs[1]): `EPSesPlusLinear(((3,4),), p=1)` on
MNIST-shaped synthetic input x (1, B, 28, 28, 2) = (sin^2, cos^2)(pi u / 2), bf16 tensors with f32
accumulation, B = 1024 per GPU.  A step follows dctn/benchmark.py:40-43: `model(x).backward(out_grad)`
with a fixed `out_grad = randn_like(out)`; with N > 1 every rank runs its own batch (weak scaling)
and the step ends with the flat-bucket RCCL all-reduce of the parameter gradients.
A window = one output site of one sample: 26*26 = 676 windows per sample.

Timing: W warm-up steps, then BLOCKS (5) blocks of EXACTLY K steps, each bracketed by a barrier and a device
synchronisation; `ms_per_step` / `value` come from the MEDIAN block (max over ranks per block) - one 0.8 ms
block moved by 20 % from run to run in round 1 - and every block's time is in `config.blocks_ms`.

Prints ONE JSON line on rank 0.  Beside the headline fields it carries `roofline` and `cpu_baseline` for the
headline workload and, at N = 1, `configs`: one entry per other BASELINE / SURVEY 8(d) configuration (cfg1 f64,
cfg3a f32, cfg3b f32, cfg4 ConvSBS r = 4 and r = 16, cfg5 logmatmulexp fold), each timed here with the protocol
of dctn/benchmark.py:14-56 and priced against its own roofline (see README / DESIGN.md section 5).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# MI355X_MICROARCH.md: dense MFMA peaks (no sparsity): bf16 ~2.5 PFLOP/s; exact f32 157.3 TFLOP/s; f64 78.6
MFMA_PEAK_TFLOPS = {"bfloat16": 2500.0, "float32": 157.3, "float64": 78.6}
DTYPE_NAME = {torch.bfloat16: "bf16", torch.float32: "f32", torch.float64: "f64"}
BLOCKS = 5

WORKLOADS = {
    # name: (epses_specs, image_size, Q0, dtype)
    "cfg2": (((3, 4),), 28, 2, torch.bfloat16),
    "cfg2_f32": (((3, 4),), 28, 2, torch.float32),
    "cfg3a": (((4, 4), (3, 6)), 28, 2, torch.float32),
    "cfg3a_bf16": (((4, 4), (3, 6)), 28, 2, torch.bfloat16),   # two-halves GEMMs on the bf16 matrix cores
    "cfg3b": (((4, 8), (2, 8)), 28, 2, torch.float32),
    "cfg3b_bf16": (((4, 8), (2, 8)), 28, 2, torch.bfloat16),
    # what the directory BASELINE configs[3] names really launches (check_super_small_model.sh:1-12: new_runner.py
    # --ds-type cifar10_ycbcr --epses-specs '(3,6)' --add-constant-channel): one EPS K=3 on 32x32, Q0 = 3 colour values + 1
    "cfg4_eps36": (((3, 6),), 32, 4, torch.float32),
}
EXTRA_CONFIGS = ("cfg2_f32", "cfg1", "cfg3a", "cfg3a_bf16", "cfg3b", "cfg4_r4", "cfg4_r8", "cfg4_r16", "cfg4_eps36", "cfg5")
# the side configurations whose three scalars (_ms, _frac, _cpu_wps) lead `config` (the driver's record keeps the first
# ~900 characters of it): every BASELINE / SURVEY 8(d) configuration; cfg3a_bf16 (no BASELINE config) stays in `configs`
# and `side_summary` only
CONFIG_SCALARS = ("cfg2_f32", "cfg1", "cfg3a", "cfg3b", "cfg4_r4", "cfg4_r8", "cfg4_r16", "cfg4_eps36", "cfg5")
# --workload also takes the two BASELINE configs that are not EPS models (configs[3] "ConvSBS ... DDP over 8xMI355X",
# configs[4] "logmatmulexp ... 8xMI355X"): the same sharding, timing protocol and JSON line as the EPS workloads
SIDE_WORKLOADS = ("cfg4_r4", "cfg4_r8", "cfg4_r16", "cfg5")
CFG5_SITES = 26 * 26   # windows per sample of cfg5: MNIST-sized window counts, 9 matrices of 16 x 16 per window
SNAKE = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]   # mnist.py:190-199


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def synthetic_input(batch, image_size, q0, dtype, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    u = torch.rand(1, batch, image_size, image_size, generator=g)
    if q0 == 2:  # dataset_loading.py:33-36,63 feature map with nu = 1
        x = torch.stack((torch.sin(u * torch.pi / 2) ** 2, torch.cos(u * torch.pi / 2) ** 2), dim=-1)
    elif q0 == 4:  # dataset_loading.py:349-364: three colour values (standardised) + the constant-1 channel
        x = torch.cat((torch.randn(1, batch, image_size, image_size, 3, generator=g),
                       torch.ones(1, batch, image_size, image_size, 1)), dim=-1)
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


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def time_cpu(step, target_seconds, max_iters=200):
    """(seconds per call, calls) of a CPU callable: one warm-up, one probe, then as many calls as fit the budget."""
    step()
    t0 = time.perf_counter()
    step()
    one = time.perf_counter() - t0
    iters = max(1, min(max_iters, int(target_seconds / max(one, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    return (time.perf_counter() - t0) / iters, iters


def cpu_baseline_eps_model(specs, image_size, q0, dtype=torch.float32, batch=128, target_seconds=12.0):
    """The oracle's restatement of the reference's 4-step path (oracle/ref_cpu.py), all host cores, same fwd+bwd
    protocol, on a bounded sample of the workload."""
    from oracle import ref_cpu as R

    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    x = synthetic_input(batch, image_size, q0, dtype, "cpu", 0)
    epses, in_size = [], q0
    for k, o in specs:
        epses.append((torch.randn(*(in_size,) * (k * k), o, dtype=dtype) * in_size ** (-k * k / 2)).requires_grad_(True))
        in_size = o
    side = image_size - sum(k for k, _ in specs) + len(specs)
    weight = (torch.randn(10, side * side * in_size, dtype=dtype) * 0.01).requires_grad_(True)
    bias = torch.zeros(10, dtype=dtype, requires_grad=True)
    out_grad = torch.randn(batch, 10, dtype=dtype)
    # layers after the first get an input gradient through autograd, the first does not (x is the dataset tensor)

    def step():
        R.eps_plus_linear_forward(epses, weight, bias, x).backward(out_grad)

    dt, iters = time_cpu(step, target_seconds)
    wps = windows_per_sample(specs, image_size) * batch / dt
    return {
        "value": wps, "unit": "windows/s", "cores": cores, "kind": "port",
        "sample": f"oracle 4-step path (torch CPU {DTYPE_NAME[dtype]}, {cores} threads, {cpu_model_name()}), batch {batch}, "
                  f"{iters} fwd+bwd iterations, {dt*1e3:.1f} ms/iteration",
    }


# ------------------------------------------------------------------------------------------ device timing
def safe_capture(body, dev, warm=2):
    """`body` captured into a HIP graph on a stream of its own, or None when the capture fails (the thread is then
    taken out of capture mode and given its stream back).  Never used for bodies that contain a collective unless
    a child-process probe has shown that such a capture works (see main)."""
    prev = torch.cuda.current_stream(dev)
    g = None
    try:
        side = torch.cuda.Stream(dev)
        side.wait_stream(prev)
        with torch.cuda.stream(side):
            for _ in range(warm):
                body()
        prev.wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=torch.cuda.Stream(dev), capture_error_mode="thread_local"):
            body()
        return g
    except Exception as e:
        log(f"HIP graph capture failed ({type(e).__name__}: {e}); timing eager launches")
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


def device_time(fn, dev, iters, graph=True, blocks=3):
    """Median over `blocks` blocks of (HIP-event time of `iters` back-to-back calls) / iters, in seconds, on the
    stream the kernels are launched on (torch's current stream).  With `graph` the call is replayed from a HIP
    graph, so host launch overhead does not count and successive calls stay dependent."""
    runner = fn
    if graph:
        g = safe_capture(fn, dev)
        if g is not None:
            runner = g.replay
    else:
        for _ in range(2):
            fn()
    runner()
    torch.cuda.synchronize(dev)
    times = []
    for _ in range(blocks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            runner()
        e1.record()
        torch.cuda.synchronize(dev)
        times.append(e0.elapsed_time(e1) / iters * 1e-3)
    return statistics.median(times)


_TRAFFIC = None
TRAFFIC_FILE = "r05_pmc_traffic.json"   # the current round's passes only (tools/profile_round.sh + condense_round.py)


def library_sha256():
    """sha256 of the kernel library this process loads (dctn_amd/libdctn_amd.so): what ties the committed PMC passes to
    the code that runs (the passes' `_meta.so_sha256` is taken on the GPU box from the library they profiled)."""
    import hashlib

    path = os.path.join(ROOT, "dctn_amd", "libdctn_amd.so")
    try:
        h = hashlib.sha256()
        with open(path, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
        return h.hexdigest()
    except OSError:
        return None


def traffic_table():
    """The committed PMC table, or {} when it was taken on ANOTHER build of the library than the one loaded now (then
    every `roofline.traffic` is null: a number that nothing ties to the running code is not reported)."""
    global _TRAFFIC
    if _TRAFFIC is None:
        _TRAFFIC = {}
        path = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        if os.path.exists(path):
            try:
                table = json.load(open(path))
            except Exception:
                table = {}
            meta = table.get("_meta", {}) if isinstance(table.get("_meta"), dict) else {}
            match = bool(meta.get("so_sha256")) and meta.get("so_sha256") == library_sha256()
            _TRAFFIC = table if match else {"_meta": meta}
            _TRAFFIC["_match"] = match
    return _TRAFFIC


def traffic_stamp():
    """(git head the passes were taken on, whether they were taken on the library that is loaded now)"""
    t = traffic_table()
    return t.get("_meta", {}).get("head"), bool(t.get("_match"))


def pmc_traffic(key):
    """HBM-side bytes per launch from the committed PMC passes (profiles/<round>_pmc_traffic.json, produced by
    tools/condense_round.py from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this script); None when the
    kernel has not been profiled or the passes belong to another build of the library."""
    table = traffic_table()
    if ":" in key and key.split(":", 1)[0] in table and isinstance(table[key.split(":", 1)[0]], list):
        # side config: "<cfg>:<kernel substring>" -> the longest-running instantiation of that kernel in the config's profile
        cfg, sub = key.split(":", 1)
        hits = [e for e in table[cfg] if sub in e["kernel"] and e.get("avg_us")]
        if not hits:
            return None
        return int(max(hits, key=lambda e: e["avg_us"])["traffic_bytes"])
    v = table.get(key)
    return int(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else None


def roofline_entry(bound, kernel, call, seconds, flops, nbytes, dtype, traffic_key=None, **extra):
    """One roofline object: `achieved` = algorithmic flops (bound mfma) or bytes (bound hbm) of the call divided by
    its measured duration."""
    if bound == "mfma":
        peak = MFMA_PEAK_TFLOPS[str(dtype).replace("torch.", "")]
        achieved, unit = flops / seconds / 1e12, "TFLOP/s"
    else:
        peak, achieved, unit = HBM_PEAK_GBS, nbytes / seconds / 1e9, "GB/s"
    head, match = traffic_stamp()
    entry = {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
             "traffic": pmc_traffic(traffic_key) if traffic_key else None, "traffic_head": head,
             "traffic_on_this_library": match, "kernel": kernel, "call": call,
             "launch_us": seconds * 1e6, "algorithmic_flops": int(flops), "algorithmic_bytes": int(nbytes),
             "hbm_gbs": nbytes / seconds / 1e9}
    entry.update(extra)
    return entry


# ------------------------------------------------------------------------------------------ EPS calls
def eps_call_timers(core, x, need_dx, dev, head=None):
    """The forward and the backward C-ABI call of ONE EPS layer as closures over preallocated buffers.
    `head` = (weight, bias) selects the fused EPS + linear-head backward where the model uses it."""
    from dctn_amd import _lib as L

    C, B, H, W, Q = x.shape
    N = core.ndim - 1
    K = int(round((N // C) ** 0.5))
    O = core.shape[-1]
    Ho = H - K + 1
    out = torch.empty((B, Ho, Ho, O), dtype=x.dtype, device=dev)
    dy = torch.randn((B, Ho, Ho, O), device=dev).to(x.dtype)
    dcore = torch.empty_like(core)
    dx = torch.empty_like(x) if need_dx else None
    lib, code, pol = L.lib(), L.dtype_code(x), L.precision()
    wsf = L.workspace(lib.dctn_eps_fwd_workspace_bytes(C, B, H, W, Q, K, O, code, pol), dev)
    wsb = L.workspace(lib.dctn_eps_bwd_workspace_bytes(C, B, H, W, Q, K, O, code, pol, int(need_dx), 1), dev).clone()

    # the training forward of a layer whose input needs a gradient keeps its GEMM result for the backward
    nsaved = lib.dctn_eps_saved_bytes(C, B, H, W, Q, K, O, code, pol) if need_dx else 0
    saved = torch.empty(nsaved, dtype=torch.uint8, device=dev) if nsaved else None
    kept = [False]

    def fwd():
        if saved is None:
            L.check(lib.dctn_eps_fwd(x.data_ptr(), L.strides5(x), core.data_ptr(), out.data_ptr(), wsf.data_ptr(),
                                     wsf.numel(), C, B, H, W, Q, K, O, code, pol, L.stream_ptr(dev)), "fwd")
        else:
            kept[0] = L.check(lib.dctn_eps_fwd_save(x.data_ptr(), L.strides5(x), core.data_ptr(), out.data_ptr(),
                                                    saved.data_ptr(), saved.numel(), wsf.data_ptr(), wsf.numel(),
                                                    C, B, H, W, Q, K, O, code, pol, L.stream_ptr(dev)), "fwd") == L.SAVED

    fused = False
    if head is not None:
        from dctn_amd.eps_plus_linear import _EpsLinearHeadFunction

        w_head, b_head = head
        fused = _EpsLinearHeadFunction.supported(core, x, w_head, b_head)
    if fused:
        cout = w_head.shape[0]
        feat = out.view(B, -1)
        dl = (torch.randn((B, cout), device=dev) * 0.1).to(x.dtype)
        dw, db = torch.empty_like(w_head), torch.empty_like(b_head)
        wsh = L.workspace(lib.dctn_eps_head_bwd_workspace_bytes(C, B, H, W, Q, K, O, cout, code, pol), dev).clone()
        logits = torch.empty((B, cout), dtype=x.dtype, device=dev)

        def head_fwd():   # layer + flatten + head as ONE kernel; ERR_UNSUPPORTED where the model composes two calls
            return lib.dctn_eps_head_fwd(x.data_ptr(), L.strides5(x), core.data_ptr(), w_head.data_ptr(), b_head.data_ptr(),
                                         feat.data_ptr(), logits.data_ptr(), C, B, H, W, Q, K, O, cout, code, pol, L.stream_ptr(dev))

        def bwd(policy=pol):
            L.check(lib.dctn_eps_head_bwd(x.data_ptr(), L.strides5(x), feat.data_ptr(), dl.data_ptr(), w_head.data_ptr(),
                                          dcore.data_ptr(), dw.data_ptr(), db.data_ptr(), wsh.data_ptr(), wsh.numel(),
                                          C, B, H, W, Q, K, O, cout, code, policy, L.stream_ptr(dev)), "head bwd")
    else:
        def bwd(policy=pol):
            tail = (None if dx is None else dx.data_ptr(), dcore.data_ptr(), wsb.data_ptr(), wsb.numel(),
                    C, B, H, W, Q, K, O, code, policy, L.stream_ptr(dev))
            if kept[0]:   # (the forward closure ran first and left `saved`)
                L.check(lib.dctn_eps_bwd_saved(x.data_ptr(), L.strides5(x), core.data_ptr(), dy.data_ptr(),
                                               saved.data_ptr(), saved.numel(), *tail), "bwd")
            else:
                L.check(lib.dctn_eps_bwd(x.data_ptr(), L.strides5(x), core.data_ptr(), dy.data_ptr(), *tail), "bwd")
    wn = B * Ho * Ho
    gemm = 2 * (Q ** N) * O   # flops per window of the core GEMM (fwd 2*Q^N*O; + the same per gradient)
    return {"fwd": fwd, "bwd": bwd, "fused": fused, "head_fwd": head_fwd if fused else None, "windows": wn, "gemm_flops": gemm, "K": K, "O": O, "N": N,
            "bytes_x": x.numel() * x.element_size(), "bytes_y": wn * O * x.element_size(),
            "bytes_core": core.numel() * core.element_size(), "saved_bytes": nsaved, "keep": (out, dy, dcore, dx, wsf, wsb, saved)}


def headline_roofline(model, x, specs, steps, ms_per_step=None):
    """The dominant KERNEL of the step that is timed.  For the register-resident families (cfg2: bf16 `eps_mfma.hip`,
    float32 `eps_q2f32.hip`) the step is three kernels - layer + head forward (`dctn_eps_head_fwd`), the dCore kernel
    and the finishing kernel of `dctn_eps_head_bwd` - each timed here as a chain of 20 dependent launches replayed from
    a HIP graph (each launch drains before the next starts, as inside the real step: the per-launch duration rocprofv3
    reports).  The dCore kernel alone is the backward call with DCTN_OPT_MAIN_KERNEL_ONLY (an explicit per-call flag
    of the C-ABI); the finishing kernel is the call's chain minus that (one kernel boundary included).  `step_frac` =
    the step's algorithmic flops / `ms_per_step` / peak."""
    from dctn_amd import _lib as L
    from dctn_amd.eps import _bf16_through_f32

    dev = x.device
    core = model.epses[0].detach().contiguous()
    if _bf16_through_f32(core, x):
        core, x = core.float(), x.float()
    w_head, b_head = model.linear.weight.detach().contiguous(), model.linear.bias.detach().contiguous()
    t = eps_call_timers(core, x, False, dev, head=(w_head, b_head) if len(specs) == 1 else None)
    C, B, H, W, Q = x.shape
    K, O, N, wn = t["K"], t["O"], t["N"], t["windows"]
    cout = w_head.shape[0]
    esz = x.element_size()
    n = max(steps, 100)
    CHAIN = 20

    def chain(fn):
        def body():
            for _ in range(CHAIN):
                fn()
        return device_time(body, dev, max(1, n // CHAIN)) / CHAIN

    t["fwd"]()
    fwd_family = L.last_kernel()
    t["bwd"]()
    bwd_family = L.last_kernel()
    reg_family = t["fused"] and ("q2reg" in bwd_family or "q2f32" in bwd_family)
    one_kernel_fwd = reg_family and t["head_fwd"]() == 0
    # this path is compute bound (MFMA for the core GEMM, VALU for the Khatri-Rao halves);
    # algorithmic flops per window: forward 2*Q^N*O (GEMM) + the two halves and the final dot (+ the head: 2*Cout*O);
    # dCore kernel the same GEMM size transposed + forming dY (2*Cout*O); dW = dLogits^T x features (2*Cout*O) belongs
    # to the finishing kernel
    half = 2 * (Q ** ((N + 1) // 2) + Q ** (N // 2)) + 2 * Q ** (N // 2) * O
    head_fl = 2 * cout * O if t["fused"] else 0
    flops = {"fwd": wn * (t["gemm_flops"] + half + head_fl), "dcore": wn * (t["gemm_flops"] + half + head_fl), "finish": wn * head_fl}
    head_bytes = (B * cout + w_head.numel()) * esz
    alg = {"fwd": t["bytes_x"] + t["bytes_y"] + t["bytes_core"] + (head_bytes if t["fused"] else 0),
           # fused: the dCore kernel reads x, dLogits and the head weight and produces dCore; the features are read and
           # dWeight / dBias written by the finishing kernel
           "dcore": (t["bytes_x"] + t["bytes_core"] + head_bytes) if t["fused"] else t["bytes_x"] + t["bytes_y"] + t["bytes_core"],
           "finish": t["bytes_y"] + head_bytes + t["bytes_core"]}
    step_flops = flops["fwd"] + flops["dcore"] + flops["finish"]
    peak = MFMA_PEAK_TFLOPS[str(x.dtype).replace("torch.", "")]
    extra = {"fused_head": bool(t["fused"]), "step_algorithmic_flops": int(step_flops), "flops_per_window_step": step_flops / wn}
    if ms_per_step:
        extra["step_tflops"] = step_flops / (ms_per_step * 1e-3) / 1e12
        extra["step_frac"] = extra["step_tflops"] / peak
    if reg_family:
        bf16 = "q2reg" in bwd_family
        names = {"fwd": ("eps_fwd_head_q2reg_t_k" if bf16 else "eps_fwd_q2f32_k") if one_kernel_fwd else fwd_family,
                 "dcore": "eps_bwd_dcore_q2reg_k" if bf16 else "eps_bwd_q2f32_k",
                 "finish": "eps_head_reduce_k" if bf16 else "eps_q2f32_finish_k"}
        main_only = L.precision() | L.OPT_MAIN_KERNEL_ONLY
        sec = {"fwd": chain(t["head_fwd"] if one_kernel_fwd else t["fwd"]), "dcore": chain(lambda: t["bwd"](main_only))}
        call_bwd = chain(t["bwd"])
        sec["finish"] = max(call_bwd - sec["dcore"], 0.0)
        dom = max(sec, key=lambda k: sec[k])
        kname = names[dom]
        extra.update(kernels_us={names[k]: v * 1e6 for k, v in sec.items()},
                     kernels_frac={names[k]: (flops[k] / v / 1e12 / peak if v > 0 else None) for k, v in sec.items() if k != "finish"},
                     calls_us={"dctn_eps_head_fwd" if one_kernel_fwd else "dctn_eps_fwd": sec["fwd"] * 1e6, "dctn_eps_head_bwd": call_bwd * 1e6},
                     finish_kernel_note="call chain minus dCore-kernel chain (one kernel boundary included)")
        call = {"fwd": "dctn_eps_head_fwd" if one_kernel_fwd else "dctn_eps_fwd", "dcore": "dctn_eps_head_bwd", "finish": "dctn_eps_head_bwd"}[dom]
        seconds = sec[dom]
    else:
        calls = {"fwd": device_time(t["fwd"], dev, n, graph=False), "dcore": device_time(t["bwd"], dev, n, graph=False)}
        dom = max(calls, key=lambda k: calls[k])
        kname, seconds = (bwd_family if dom == "dcore" else fwd_family), calls[dom]
        call = "dctn_eps_fwd" if dom == "fwd" else ("dctn_eps_head_bwd" if t["fused"] else "dctn_eps_bwd")
        extra.update(calls_us={k: v * 1e6 for k, v in calls.items()})
    return roofline_entry("mfma", kname, call, seconds, flops[dom], alg[dom], x.dtype, traffic_key=f"{kname}:B{B}",
                          flops_per_window=flops[dom] / wn, bytes_per_window=alg[dom] / wn,
                          hbm_frac=alg[dom] / seconds / 1e9 / HBM_PEAK_GBS, **extra)


def convsbs_call_timers(string, x, dy, dev):
    """Forward and backward C-ABI calls of one ConvSBS string over preallocated buffers (no allocation or autograd
    bookkeeping inside the timed region: the r = 4 forward kernel is shorter than a Python-side launch)."""
    from dctn_amd import _lib as L

    spec = string.spec
    C, B, H, W, q = x.shape
    n, shapes = len(spec), spec.shapes
    cores = [c.detach().contiguous() for c in string.cores]
    outs = L.int_array([s_.out_quantum_dim_size for s_ in shapes])
    bonds = L.int_array(spec.bond_sizes)
    ph, pw = L.int_array([p.h for p in spec.positions]), L.int_array([p.w for p in spec.positions])
    out = torch.empty((B, H - spec.max_height_pos, W - spec.max_width_pos, spec.out_total_quantum_dim_size), dtype=x.dtype, device=dev)
    lib, code = L.lib(), L.dtype_code(x)
    # the training forward: its workspace is the buffer the forward states are left in for dctn_convsbs_bwd_saved
    saved = lib.dctn_convsbs_saved_states_bytes(n, outs, bonds, C, B, H, W, q, ph, pw, code)
    wsf = torch.empty(max(256, saved, lib.dctn_convsbs_workspace_bytes(n, outs, bonds, C, B, H, W, q, ph, pw, code, 0)), dtype=torch.uint8, device=dev)
    wsb = torch.empty(max(256, lib.dctn_convsbs_workspace_bytes(n, outs, bonds, C, B, H, W, q, ph, pw, code, 1)), dtype=torch.uint8, device=dev)
    dx = torch.empty_like(x)
    flat = torch.empty(sum(c.numel() for c in cores), dtype=x.dtype, device=dev)
    dcores, off = [], 0
    for c in cores:
        dcores.append(flat[off:off + c.numel()].view_as(c))
        off += c.numel()
    cp, dcp = L.ptr_array(cores), L.ptr_array(dcores)
    xs = L.strides5(x)

    def fwd():
        L.check(lib.dctn_convsbs_fwd(x.data_ptr(), xs, cp, out.data_ptr(), n, outs, bonds, ph, pw, C, B, H, W, q,
                                     wsf.data_ptr(), wsf.numel(), code, L.stream_ptr(dev)), "convsbs fwd")

    def bwd():   # (after fwd(): the states it left are taken over; saved == 0: the backward recomputes them)
        L.check(lib.dctn_convsbs_bwd_saved(x.data_ptr(), xs, cp, dy.data_ptr(), dx.data_ptr(), dcp, n, outs, bonds, ph, pw,
                                           C, B, H, W, q, wsb.data_ptr(), wsb.numel(), wsf.data_ptr() if saved else None,
                                           saved, code, L.stream_ptr(dev)), "convsbs bwd")

    return fwd, bwd, (cores, out, wsf, wsb, dx, flat, dcores)


# ------------------------------------------------------------------------------------------ other configs
def extra_eps_model(name, dev, iters):
    """cfg3a / cfg3b: the two-EPS model in the reference's own arithmetic (float32), B = 128: the whole
    `model(x).backward(out_grad)` step, and each layer's forward / backward call on its own."""
    from dctn_amd import _lib as L
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd

    specs, image_size, q0, dtype = WORKLOADS[name]
    batch = 128
    torch.manual_seed(0)
    model = EPSesPlusLinear(specs, UnitTheoreticalOutputStd(), 1.0, dev, dtype, image_size=image_size, Q_0=q0)
    x = synthetic_input(batch, image_size, q0, dtype, dev, seed=1)
    out_grad = torch.randn(batch, 10, device=dev).to(dtype)

    def fwd():
        with torch.no_grad():
            model(x)

    def fwd_bwd():
        for p in model.parameters():
            p.grad = None
        model(x).backward(out_grad)

    t_f = device_time(fwd, dev, iters, graph=False)
    t_fb = device_time(fwd_bwd, dev, iters, graph=False)
    # per layer: forward call, backward call (layer 1: dCore only; deeper layers: dX too)
    layers, xin = [], x
    for li, (core, (k, o)) in enumerate(zip(model.epses, specs)):
        c = core.detach().contiguous()
        t = eps_call_timers(c, xin, li > 0, dev)
        sf = device_time(t["fwd"], dev, max(3, iters // 2), graph=False)
        t["fwd"]()
        kf = L.last_kernel()
        sb = device_time(t["bwd"], dev, max(3, iters // 2), graph=False)
        t["bwd"]()
        kb = L.last_kernel()
        nb = 2 if li > 0 else 1
        layers.append({"layer": li + 1, "spec": [k, o], "windows": t["windows"],
                       "fwd_us": sf * 1e6, "fwd_family": kf, "fwd_tflops": t["windows"] * t["gemm_flops"] / sf / 1e12,
                       "bwd_us": sb * 1e6, "bwd_family": kb, "bwd_tflops": nb * t["windows"] * t["gemm_flops"] / sb / 1e12,
                       "_t": t, "_nb": nb})
        with torch.no_grad():
            from dctn_amd.eps import eps
            xin = eps(c, xin).unsqueeze(0)
    windows = windows_per_sample(specs, image_size) * batch
    step_flops = sum(l["windows"] * l["_t"]["gemm_flops"] * (1 + l["_nb"]) for l in layers)
    # dominant call of the step
    cand = []
    for l in layers:
        t = l["_t"]
        cand.append((l["fwd_us"], f"L{l['layer']} forward", "dctn_eps_fwd", l["fwd_family"], l["windows"] * t["gemm_flops"],
                     t["bytes_x"] + t["bytes_y"] + t["bytes_core"]))
        cand.append((l["bwd_us"], f"L{l['layer']} backward", "dctn_eps_bwd", l["bwd_family"],
                     l["_nb"] * l["windows"] * t["gemm_flops"],
                     t["bytes_x"] * l["_nb"] + t["bytes_y"] + t["bytes_core"] * 2))
    us, what, call, fam, fl, by = max(cand)
    kernel = {"eps_fwd_mfma_bigcore_f32": "eps_bigcore_k",
              "eps_bwd_mfma_bigcore_f32": "eps_bigcore_k (G0, G1) + eps_bigcore_dcore_k" if "L1" not in what else "eps_bigcore_dcore_k",
              "eps_bwd_mfma_bigcore_f32_savedz": "eps_bigcore_k (G0) + eps_bigcore_dp1_k (saved Z) + eps_bigcore_dcore_k"}.get(fam, fam)
    roof = roofline_entry("mfma", kernel, f"{call} ({what})", us * 1e-6, fl, by, dtype, traffic_key=f"{name}:eps_bigcore_k",
                          traffic_scope=("longest eps_bigcore_k instantiation of the profiled config (profiles/" + TRAFFIC_FILE + ")"
                                         if dtype == torch.float32 else "not collected for this config"),
                          step_tflops=step_flops / t_fb / 1e12, step_frac=step_flops / t_fb / 1e12 / MFMA_PEAK_TFLOPS[str(dtype).replace("torch.", "")],
                          step_algorithmic_flops=int(step_flops))
    for l in layers:
        del l["_t"], l["_nb"]
    arith = ("f32" if dtype == torch.float32 else
             "bf16 storage, bf16 matrix cores with float32 accumulation (not a BASELINE config: the cfg3a model under the bf16 policy)")
    layout = ("CIFAR YCbCr + constant channel layout 32x32 Q0=4 (check_super_small_model.sh:1-12)" if q0 == 4
              else f"MNIST-shaped {image_size}x{image_size} Q0={q0}")
    return {"workload": f"{name}: EPSesPlusLinear({specs}) {arith} on {layout}, fwd + bwd(out_grad), batch {batch}",
            "dtype": DTYPE_NAME[dtype], "windows_per_step": windows, "ms_per_step": t_fb * 1e3, "fwd_ms": t_f * 1e3,
            "value": windows / t_fb, "unit": "windows/s", "layers": layers, "roofline": roof,
            "cpu_baseline": cpu_baseline_eps_model(specs, image_size, q0, torch.float32 if dtype == torch.bfloat16 else dtype,
                                                   batch=8, target_seconds=4.0)}


def extra_cfg2_f32(dev, batch=1024, gsteps=20):
    """cfg2 in the reference's OWN arithmetic (new_runner.py:417 runs float32 and nothing else): the headline model,
    float32 tensors, B = 1024, the headline's protocol (fwd + bwd(out_grad) replayed from a HIP graph of 20 steps) and
    the headline's roofline object (the step's three kernels timed as chains of dependent launches)."""
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd

    specs, image_size, q0, dtype = WORKLOADS["cfg2_f32"]
    torch.manual_seed(0)
    model = EPSesPlusLinear(specs, UnitTheoreticalOutputStd(), 1.0, dev, dtype, image_size=image_size, Q_0=q0)
    params = list(model.parameters())
    x = synthetic_input(batch, image_size, q0, dtype, dev, seed=1)
    out_grad = torch.randn(batch, 10, device=dev).to(dtype)

    def fwd():
        with torch.no_grad():
            model(x)

    def fwd_bwd():
        for p in params:
            p.grad = None
        model(x).backward(out_grad)

    def several(fn):
        def body():
            for _ in range(gsteps):
                fn()
        return body

    t_fb = device_time(several(fwd_bwd), dev, 10, blocks=5) / gsteps
    t_f = device_time(several(fwd), dev, 10) / gsteps
    windows = windows_per_sample(specs, image_size) * batch
    roof = headline_roofline(model, x, specs, 100, ms_per_step=t_fb * 1e3)
    return {"workload": f"cfg2_f32: EPSesPlusLinear({specs}) float32 (the reference's dtype) on MNIST-shaped 28x28 Q0=2, fwd + bwd(out_grad), "
                        f"batch {batch}, HIP graph of {gsteps} steps",
            "dtype": "f32", "windows_per_step": windows, "ms_per_step": t_fb * 1e3, "fwd_ms": t_f * 1e3,
            "value": windows / t_fb, "unit": "windows/s", "roofline": roof,
            "cpu_baseline": cpu_baseline_eps_model(specs, image_size, q0, torch.float32, batch=64, target_seconds=4.0)}


def extra_cfg1(dev, iters):
    """cfg1: the reference's own EPS micro-benchmark (small_experiments/eps2d_benchmark/benchmark.py:45-64):
    eps(core, x), B = 64, 28x28, K = 4, Q = 2, O = 2, float64, everything randn, core AND input require grad."""
    from dctn_amd import _lib as L
    from dctn_amd.eps import eps
    from oracle import ref_cpu as R

    B, HW, Q, K, O, dt = 64, 28, 2, 4, 2, torch.float64
    N = K * K
    torch.manual_seed(0)
    x = torch.randn(1, B, HW, HW, Q, device=dev, dtype=dt, requires_grad=True)
    core = torch.randn(*(Q,) * N, O, device=dev, dtype=dt, requires_grad=True)
    y = eps(core, x)
    dy = torch.randn_like(y)
    windows = y.shape[0] * y.shape[1] * y.shape[2]

    def fwd():
        with torch.no_grad():
            eps(core, x)

    def fwd_bwd():
        x.grad = None
        core.grad = None
        eps(core, x).backward(dy)

    t_f, t_fb = device_time(fwd, dev, iters, graph=False), device_time(fwd_bwd, dev, iters, graph=False)
    fwd()
    fam = L.last_kernel()
    t = eps_call_timers(core.detach(), x.detach(), True, dev)
    sf = device_time(t["fwd"], dev, iters, graph=False)
    sb = device_time(t["bwd"], dev, iters, graph=False)
    gemm = t["gemm_flops"] * windows
    by_b = 2 * t["bytes_x"] + t["bytes_y"] + 2 * t["bytes_core"]
    roof = roofline_entry("mfma", "halves_gemm_k<double> (3 GEMMs: Z in the forward - kept -, dCore and dP0 in the backward)", "dctn_eps_bwd", sb, 2 * gemm, by_b, dt,
                          traffic_key="cfg1:halves_gemm_k", traffic_scope="longest halves_gemm_k instantiation (one of the four GEMMs of the call)", fwd_call_us=sf * 1e6, fwd_tflops=gemm / sf / 1e12,
                          step_tflops=3 * gemm / t_fb / 1e12, step_frac=3 * gemm / t_fb / 1e12 / MFMA_PEAK_TFLOPS["float64"],
                          family=fam,
                          # measured, not a peak to price against: v_mfma_f64_16x16x4_f64 alone (4 accumulators, 4 waves per
                          # SIMD, nothing else in the loop) reaches 47.7 of the nominal 78.6 TFLOP/s on MI355X
                          instruction_ceiling_tflops=47.7, instruction_ceiling_source="tools/probes/mfma_f64_rate.hip",
                          frac_of_instruction_ceiling=2 * gemm / sb / 1e12 / 47.7)
    # CPU: the oracle's 4-step path in float64 on a bounded sample (8 of the 64 samples)
    cores_n = usable_cores()
    torch.set_num_threads(cores_n)
    Bc = 8
    xc = x.detach().cpu()[:, :Bc].clone().requires_grad_(True)
    cc = core.detach().cpu().clone().requires_grad_(True)
    dyc = dy.cpu()[:Bc]
    dtc, it = time_cpu(lambda: R.eps_4step(cc, xc).backward(dyc), 4.0)
    return {"workload": "cfg1: eps(core, x) B=64 C=1 28x28 K=4 Q=2 O=2 float64, randn core and input, fwd + bwd(dX, dCore)",
            "dtype": "f64", "windows_per_step": windows, "ms_per_step": t_fb * 1e3, "fwd_ms": t_f * 1e3,
            "value": windows / t_fb, "unit": "windows/s", "roofline": roof,
            "cpu_baseline": {"value": windows * Bc / B / dtc, "unit": "windows/s", "cores": cores_n, "kind": "port",
                             "sample": f"oracle 4-step path (torch CPU f64, {cores_n} threads, {cpu_model_name()}), batch {Bc} of {B}, "
                                       f"{it} fwd+bwd iterations, {dtc*1e3:.1f} ms/iteration"}}


def extra_cfg4(r, dev, iters):
    """cfg4: ConvSBS, the 9-core snake of mnist.py:190-199 (outs 1,1,1,1,2,1,1,1,1; open chain, bond r) on the
    CIFAR colour layout x (1, 128, 32, 32, 3) (dataset_loading.py:192-196), float32, fwd + bwd (dX and dCores)."""
    import dctn_amd
    from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
    from dctn_amd.conv_sbs_spec import SBSSpecCore
    from dctn_amd.pos2d import Pos2D
    from oracle import ref_cpu as R

    C, q, HW, B = 1, 3, 32, 128
    spec = (tuple(SBSSpecCore(Pos2D(*p), 2 if i == 4 else 1) for i, p in enumerate(SNAKE)),)
    torch.manual_seed(r)
    many = ManyConvSBS(C, q, r, False, spec, (DumbNormalInitialization((q ** C * r) ** -0.5),)).to(dev)
    x = torch.randn(C, B, HW, HW, q, device=dev, requires_grad=True)
    with torch.no_grad():   # (no autograd graph may outlive this line: its AccumulateGrad nodes would tie the later
        (y,) = many(x)      # HIP-graph capture of the step to the stream of this eager call)
    dy = torch.randn_like(y)
    windows = y.shape[0] * y.shape[1] * y.shape[2]

    def fwd():
        with torch.no_grad():
            many(x)

    def fwd_bwd():
        x.grad = None
        for prm in many.parameters():
            prm.grad = None
        many(x)[0].backward(dy)

    # The step is the nn.Module's forward + backward (the drop-in surface), replayed from a HIP graph like the headline
    # (4 steps per launch; eagerly the r = 4 step is bound by the host's autograd bookkeeping: `module_eager_ms`);
    # the two C-ABI calls on their own give the per-call durations the roofline prices
    t_f_mod, t_fb_eager = device_time(fwd, dev, iters, graph=False), device_time(fwd_bwd, dev, iters, graph=False)

    def four_steps():
        for _ in range(4):
            fwd_bwd()

    t_fb = device_time(four_steps, dev, max(4, iters // 2), graph=True) / 4
    fwd_bwd()
    fam = dctn_amd.last_kernel()
    c_fwd, c_bwd, _keep = convsbs_call_timers(many.strings[0], x.detach(), dy, dev)
    t_f = device_time(c_fwd, dev, 5 * iters, graph=False)
    t_b = device_time(c_bwd, dev, 2 * iters, graph=False)
    # algorithmic work per window in the reference's order: step A 2*q^C*sum(o*l*r), then the chain
    shapes = many.strings[0].spec.shapes
    step_a = 2 * q ** C * sum(s.out_quantum_dim_size * s.bond_left_size * s.bond_right_size for s in shapes)
    chain, oacc = 0, shapes[0].out_quantum_dim_size
    for s in shapes[1:]:
        chain += 2 * oacc * s.bond_left_size * s.out_quantum_dim_size * s.bond_right_size
        oacc *= s.out_quantum_dim_size
    flops_fwd = (step_a + chain) * windows
    n_par = sum(p.numel() for p in many.parameters())
    by_fwd = x.numel() * 4 + y.numel() * 4 + n_par * 4
    by_fb = 3 * x.numel() * 4 + 2 * y.numel() * 4 + 3 * n_par * 4     # + read x again, read dY, write dX, cores + dCores
    if "band" in fam:
        kname, call, tkey = ("convsbs_bwd_band_k (+ convsbs_band_tail_k)",
                             "dctn_convsbs_bwd (band-owning backward: chain recomputed in registers, nothing kept by the forward)",
                             "convsbs_bwd_band")
    elif "mfma" in fam:
        kname, call, tkey = "convsbs_bwd_mfma16_k", "dctn_convsbs_bwd_saved (forward states from dctn_convsbs_fwd)", "convsbs_bwd_mfma"
    else:
        kname, call, tkey = ("convsbs_bwd_regu_k (+ convsbs_regu_tail_k)",
                             "dctn_convsbs_bwd (register-resident sweep: nothing kept by the forward)", "convsbs_bwd_reg")
    if r >= 8:    # compute bound: the shared-core GEMMs of the sweep on the f32 matrix cores
        roof = roofline_entry("mfma", kname, call, t_b, 2 * flops_fwd, by_fb - by_fwd, torch.float32, traffic_key=f"cfg4_r{r}:{tkey}",
                              fwd_us=t_f * 1e6, fwd_tflops=flops_fwd / t_f / 1e12, step_tflops=3 * flops_fwd / t_fb / 1e12,
                              step_frac=3 * flops_fwd / t_fb / 1e12 / MFMA_PEAK_TFLOPS["float32"], family=fam)
    else:         # HBM / launch-latency bound: bytes of the fused ideal against the HBM peak
        roof = roofline_entry("hbm", kname, call, t_b, 2 * flops_fwd, by_fb - by_fwd, torch.float32, traffic_key=f"cfg4_r{r}:{tkey}",
                              fwd_us=t_f * 1e6, fwd_gbs=by_fwd / t_f / 1e9, step_gbs=by_fb / t_fb / 1e9,
                              step_frac=by_fb / t_fb / 1e9 / HBM_PEAK_GBS, family=fam)
    cores_n = usable_cores()
    torch.set_num_threads(cores_n)
    Bc = 16 if r <= 8 else 8
    cc = [c.detach().cpu().clone().requires_grad_(True) for c in many.strings[0].cores]
    xc = x.detach().cpu()[:, :Bc].clone().requires_grad_(True)
    dyc = dy.cpu()[:Bc]
    dtc, it = time_cpu(lambda: R.convsbs_forward(cc, SNAKE, xc).backward(dyc), 3.0)
    return {"workload": f"cfg4_r{r}: ConvSBS 9-core snake (mnist.py:190-199), open chain bond {r}, x (1,{B},32,32,3) CIFAR colour "
                        "layout, float32, fwd + bwd(dX, dCores)",
            "dtype": "f32", "windows_per_step": windows, "ms_per_step": t_fb * 1e3, "fwd_ms": t_f * 1e3,
            "module_fwd_ms": t_f_mod * 1e3, "module_eager_ms": t_fb_eager * 1e3, "calls_ms": (t_f + t_b) * 1e3,
            "timing": "nn.Module forward + backward replayed from a HIP graph (4 steps per launch), HIP events",
            "value": windows / t_fb, "unit": "windows/s", "roofline": roof,
            "cpu_baseline": {"value": windows * Bc / B / dtc, "unit": "windows/s", "cores": cores_n, "kind": "port",
                             "sample": f"oracle step A + chain (torch CPU f32, {cores_n} threads, {cpu_model_name()}), batch {Bc} of {B}, "
                                       f"{it} fwd+bwd iterations, {dtc*1e3:.1f} ms/iteration"}}


def extra_cfg5(dev, iters):
    """cfg5: batched logmatmulexp - per window the left fold of 9 log-matrices (16 x 16), float32, 692 224 windows
    (batch 1024 of 26 x 26 sites), the benchmark loop of small_experiments/logmatmulexp_benchmark/benchmark.py:30."""
    import dctn_amd
    from dctn_amd.logmatmulexp import logmatmulexp_fold
    from oracle import ref_cpu as R

    Wn, Ln, D = 692224, 9, 16
    torch.manual_seed(0)
    m = torch.randn(Wn, Ln, D, D, device=dev, requires_grad=True)
    y = logmatmulexp_fold(m)
    dy = torch.randn_like(y)
    del y

    def fwd():
        with torch.no_grad():
            logmatmulexp_fold(m)

    def fwd_bwd():
        m.grad = None
        logmatmulexp_fold(m).backward(dy)

    t_f = device_time(fwd, dev, iters, graph=False)
    fwd()
    fam_f = dctn_amd.last_kernel()
    t_fb = device_time(fwd_bwd, dev, max(2, iters // 2), graph=False)
    fam_b = dctn_amd.last_kernel()
    by_f = Wn * (Ln * D * D * 4 + D * D * 4)                  # read 9 matrices, write one: 10 240 B per window
    by_b = Wn * (2 * Ln * D * D * 4 + D * D * 4)              # read them again + dOut, write 9 gradients: 19 456 B
    t_b = max(t_fb - t_f, 1e-9)
    fl = Wn * (Ln - 1) * 2 * D ** 3
    roof = roofline_entry("hbm", "lme_fold16_bwd_mfma_k" if "mfma16" in fam_b else fam_b, "dctn_logmatmulexp_fold_bwd", t_b, 2 * fl, by_b,
                          torch.float32, traffic_key="cfg5:lme_fold16_bwd_mfma_k", fwd_kernel="lme_fold16_fwd_k" if "mfma16" in fam_f else fam_f,
                          fwd_us=t_f * 1e6, fwd_gbs=by_f / t_f / 1e9, fwd_frac=by_f / t_f / 1e9 / HBM_PEAK_GBS,
                          step_gbs=(by_f + by_b) / t_fb / 1e9, step_frac=(by_f + by_b) / t_fb / 1e9 / HBM_PEAK_GBS,
                          bytes_per_window={"fwd": by_f // Wn, "bwd": by_b // Wn}, formulation="factored (exp -> MFMA), prefix carried in scaled form; a window with a rejected step restarts in the log domain")
    del m
    torch.cuda.empty_cache()
    cores_n = usable_cores()
    torch.set_num_threads(cores_n)
    Wc = 4096
    mc = torch.randn(Wc, Ln, D, D, requires_grad=True)
    dyc = torch.randn(Wc, D, D)
    dtc, it = time_cpu(lambda: R.logmatmulexp_fold_batched(mc).backward(dyc), 3.0)
    return {"workload": f"cfg5: per-window left fold of {Ln} log-matrices ({D}x{D}) (logmatmulexp), float32, {Wn} windows, fwd + bwd",
            "dtype": "f32", "windows_per_step": Wn, "ms_per_step": t_fb * 1e3, "fwd_ms": t_f * 1e3,
            "value": Wn / t_fb, "unit": "windows/s", "roofline": roof,
            "cpu_baseline": {"value": Wc / dtc, "unit": "windows/s", "cores": cores_n, "kind": "port",
                             "sample": f"oracle broadcast-add + logsumexp fold (torch CPU f32, {cores_n} threads, {cpu_model_name()}), "
                                       f"{Wc} of {Wn} windows, {it} fwd+bwd iterations, {dtc*1e3:.1f} ms/iteration"}}


def _sig(v, digits=4):
    """`v` rounded to `digits` significant digits (short in the JSON line, exact enough for a record)."""
    return float(f"{v:.{digits}g}")


def side_scalars(name, e):
    """One side configuration as THREE scalar `config` keys (a driver that keeps the contract's keys keeps scalars only,
    and only the first ~900 characters of `config`): side_<cfg>_ms = ms per fwd+bwd step, _frac = the dominant call's
    fraction of its roofline, _cpu_wps = the CPU oracle's windows/s.  Everything else is in `configs` / `side_summary`."""
    if "value" not in e:
        return {f"side_{name}_error": str(e.get("error", "failed"))[:60]}
    out = {f"side_{name}_ms": _sig(e["ms_per_step"]), f"side_{name}_frac": round(e.get("roofline", {}).get("frac", 0.0), 3)}
    if "cpu_baseline" in e:
        out[f"side_{name}_cpu_wps"] = int(_sig(e["cpu_baseline"]["value"], 3))
    return out


def compact_config(base, names, entries):
    """`config` of the line: the workload keys, then three scalars per configuration of CONFIG_SCALARS."""
    cfg = dict(base)
    for name, e in zip(names, entries):
        if name in CONFIG_SCALARS:
            cfg.update(side_scalars(name, e))
    return cfg


def side_summary_row(e):
    """[ms per step, windows/s, frac, bound, step_frac, traffic / algorithmic bytes, cpu windows/s] of one side configuration."""
    if "value" not in e:
        return {"error": str(e.get("error", "failed"))[:120]}
    roof = e.get("roofline", {})
    tx = (round(roof["traffic"] / roof["algorithmic_bytes"], 2) if roof.get("traffic") is not None and roof.get("algorithmic_bytes") else None)
    return [_sig(e["ms_per_step"], 5), round(e["value"]), round(roof.get("frac", 0.0), 4), roof.get("bound"),
            round(roof["step_frac"], 4) if roof.get("step_frac") is not None else None, tx,
            round(e["cpu_baseline"]["value"]) if "cpu_baseline" in e else None]


def run_extra(name, dev):
    if name == "cfg2_f32":
        return extra_cfg2_f32(dev)
    if name == "cfg1":
        return extra_cfg1(dev, 20)
    if name in ("cfg3a", "cfg3b", "cfg3a_bf16", "cfg3b_bf16", "cfg4_eps36"):
        return extra_eps_model(name, dev, 10 if name == "cfg3a" else 20)
    if name.startswith("cfg4_r"):
        return extra_cfg4(int(name[6:]), dev, 30)
    if name == "cfg5":
        return extra_cfg5(dev, 6)
    raise SystemExit(f"unknown config {name}")


def run_extra_in_child(name, no_cpu_baseline, timeout=420.0, device_index=0):
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--skip-headline", "--configs", name, "--device", str(int(device_index or 0))]
    if no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_") and k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=sys.stderr, timeout=timeout, text=True)
    except subprocess.TimeoutExpired:
        return {"workload": name, "error": f"timed out after {timeout:.0f} s"}
    if res.returncode != 0:
        return {"workload": name, "error": f"child exited with code {res.returncode}"}
    try:
        return json.loads(res.stdout.strip().splitlines()[-1])["configs"][0]
    except Exception as e:   # noqa: BLE001
        return {"workload": name, "error": f"unparsable child output ({type(e).__name__}: {e})"}


def launch_ranks(n, argv, timeout=1500.0):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <argv>` as a CHILD process of a parent
    that has made no GPU call: stdout (rank 0's one JSON line) is relayed, stderr passes through; returns the exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:   # a free rendezvous port on the loopback address
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log(f"--gpus {n} without WORLD_SIZE: starting the ranks with torch.distributed.run on port {port}")
    child = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=sys.stderr, text=True)
    try:
        out, _ = child.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        child.kill()   # this exact child, by handle
        out, _ = child.communicate()
        log(f"the {n}-rank run did not finish within {timeout:.0f} s")
        return 124
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    return child.returncode


# ------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS) + list(SIDE_WORKLOADS))
    ap.add_argument("--batch", type=int, default=None,
                    help="samples per GPU (default 1024 for cfg2 and cfg5, 128 for cfg3 and cfg4)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --batch samples per GPU whatever N; strong: --batch is the GLOBAL batch, split over the N GPUs")
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a captured HIP graph")
    ap.add_argument("--graph-steps", type=int, default=20,
                    help="steps captured back to back into ONE HIP graph (default 20; when --steps is not a "
                         "multiple: the largest divisor of --steps below it is used).  Every step does its full work; what is amortised is the ~6-8 us the GPU's command "
                         "processor needs between two graph launches (rocprofv3: the gap between the last kernel of one "
                         "replay and the first of the next), which was a sixth of a 36 us step")
    ap.add_argument("--graph-allreduce", default="auto", choices=["auto", "0", "1"],
                    help="capture the gradient all-reduce into the step's HIP graph.  auto (default): yes over RCCL when a "
                         "child-process probe shows that the capture works on this machine; 1: the same, and fail if it does "
                         "not; 0: replay the fwd+bwd graph and launch the collective eagerly")
    ap.add_argument("--allreduce", default="rccl", choices=["rccl", "direct"],
                    help="gradient all-reduce of the step: rccl (default: torch.distributed all_reduce = RCCL ring / tree) or direct "
                         "(dctn_ar_*: one kernel per step and rank reading every peer's buffer over its own xGMI link, "
                         "always inside the step's HIP graph); the line carries allreduce_us of both")
    ap.add_argument("--time-other-allreduce", type=int, default=0,
                    help="1: at N > 1 also time the step's message through the algorithm the step does NOT use.  Off by default: "
                         "a default `--allreduce rccl` run then never creates the direct all-reduce's IPC-mapped blocks (that "
                         "path has not run across xGMI from this build; a default multi-GPU benchmark must not depend on it)")
    ap.add_argument("--configs", default="all",
                    help="the other BASELINE configs measured into `configs` at N = 1: all (default), none, or a comma list of "
                         + ", ".join(EXTRA_CONFIGS))
    ap.add_argument("--skip-headline", action="store_true", help="profiling aid: run only --configs and print their entries")
    ap.add_argument("--device", type=int, default=0, help="GPU index of a --skip-headline run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rccl-proto", default=None, help="sets NCCL_PROTO (LL / LL128 / Simple) before the communicator comes up")
    ap.add_argument("--rccl-algo", default=None, help="sets NCCL_ALGO (Ring / Tree)")
    ap.add_argument("--rccl-min-nchannels", default=None, help="sets NCCL_MIN_NCHANNELS")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); 'gloo' + "
                    "DCTN_BENCH_ONE_DEVICE=1 rehearses the multi-rank path on a single GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.skip_headline:
        # `python bench.py --gpus N` started as ONE process: start the N ranks as a child (before anything here touches
        # the GPU; this process never re-executes itself), relay rank 0's JSON line and leave with the child's code
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    from dctn_amd import ddp
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd
    import dctn_amd

    for flag, var in ((args.rccl_proto, "NCCL_PROTO"), (args.rccl_algo, "NCCL_ALGO"), (args.rccl_min_nchannels, "NCCL_MIN_NCHANNELS")):
        if flag is not None:
            os.environ[var] = str(flag)
    rccl_env = {k: os.environ.get(k) for k in ("NCCL_PROTO", "NCCL_ALGO", "NCCL_MIN_NCHANNELS")}

    if args.skip_headline:
        dev = torch.device("cuda", args.device)
        torch.cuda.set_device(dev)
        names = EXTRA_CONFIGS if args.configs == "all" else tuple(n for n in args.configs.split(",") if n)
        out = []
        for name in names:
            entry = run_extra(name, dev)
            if args.no_cpu_baseline:
                entry.pop("cpu_baseline", None)
            out.append(entry)
        print(json.dumps({"configs": out}), flush=True)
        return

    one_device = os.environ.get("DCTN_BENCH_ONE_DEVICE") == "1"
    # DCTN_BENCH_FORCE_ALLREDUCE=1: create the process group and issue the gradient all-reduce even with
    # one rank (rehearsal of the RCCL calls on a one-GPU machine; not a measurement)
    force_reduce = os.environ.get("DCTN_BENCH_FORCE_ALLREDUCE") == "1"
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    backend_name = args.backend or "nccl"
    will_reduce = env_world > 1 or force_reduce

    # Probe, in child processes and before this process forms its group, whether the collective can be captured
    # into a HIP graph here (all ranks do this at the same point; the children form a group of their own).
    probe_ok = None
    if will_reduce and args.graph and args.graph_allreduce != "0" and backend_name == "nccl":
        t0 = time.perf_counter()
        probe_ok = ddp.probe_allreduce_capture()
        log(f"all-reduce capture probe: {'ok' if probe_ok else 'FAILED'} ({time.perf_counter() - t0:.1f} s)")

    # RCCL prints a version banner on stdout when its communicator comes up; stdout must carry the one JSON line
    # only, so file descriptor 1 points at stderr until the first collective has run
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        rank, local_rank, world = ddp.init_from_env(args.backend, single_rank_group=force_reduce)
        if dist.is_initialized():
            dev0 = torch.device("cuda", 0 if one_device else local_rank)
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
    eps_model = args.workload in WORKLOADS
    batch = args.batch or (1024 if args.workload.startswith("cfg2") or args.workload == "cfg5" else 128)
    if args.scaling == "strong":
        batch = max(1, batch // world)

    torch.manual_seed(0)
    specs = image_size = q0 = model = None
    if eps_model:
        specs, image_size, q0, dtype = WORKLOADS[args.workload]
        model = EPSesPlusLinear(specs, UnitTheoreticalOutputStd(), 1.0, dev, dtype, image_size=image_size, Q_0=q0)
        params = list(model.parameters())
        ddp.broadcast_parameters(params)
        x = synthetic_input(batch, image_size, q0, dtype, dev, seed=1 + rank)  # resident in HBM before timing
        out_grad = torch.randn(batch, 10, device=dev).to(dtype)
        windows_rank = windows_per_sample(specs, image_size) * batch
        what = (f"{args.workload}: EPSesPlusLinear({specs}) on MNIST-shaped {image_size}x{image_size} Q0={q0}, "
                f"fwd + bwd(out_grad), batch {batch}/GPU")

        def fwd_bwd():
            for p in params:
                p.grad = None
            model(x).backward(out_grad)
    elif args.workload.startswith("cfg4_r"):
        # BASELINE configs[3]: the ConvSBS 9-core snake (mnist.py:190-199) on the CIFAR colour layout, batch sharded,
        # the string's flat core-gradient buffer all-reduced in place (one RCCL launch per step)
        from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
        from dctn_amd.conv_sbs_spec import SBSSpecCore
        from dctn_amd.pos2d import Pos2D

        r, dtype = int(args.workload[6:]), torch.float32
        spec = (tuple(SBSSpecCore(Pos2D(*p), 2 if i == 4 else 1) for i, p in enumerate(SNAKE)),)
        torch.manual_seed(r)
        many = ManyConvSBS(1, 3, r, False, spec, (DumbNormalInitialization((3 * r) ** -0.5),)).to(dev)
        params = list(many.parameters())
        ddp.broadcast_parameters(params)
        x = torch.randn(1, batch, 32, 32, 3, device=dev, generator=torch.Generator(dev).manual_seed(1 + rank)).requires_grad_(True)
        out_grad = torch.randn(batch, 30, 30, 2, device=dev)
        windows_rank = batch * 30 * 30
        what = (f"{args.workload}: ConvSBS 9-core snake (mnist.py:190-199), open chain bond {r}, x (1,{batch},32,32,3) CIFAR colour "
                f"layout, float32, fwd + bwd(dX, dCores) through the nn.Module, batch {batch}/GPU")

        def fwd_bwd():
            x.grad = None
            for p in params:
                p.grad = None
            many(x)[0].backward(out_grad)
    else:
        # BASELINE configs[4]: per window the left fold of 9 log-matrices (16 x 16); windows sharded over the ranks.  The
        # fold has no parameters (the reference's logmatmulexp is wired into no model: SURVEY section 0), so there is no
        # gradient to exchange: no data-path collective, N independent shards
        from dctn_amd.logmatmulexp import logmatmulexp_fold

        dtype = torch.float32
        windows_rank = batch * CFG5_SITES
        x = torch.randn(windows_rank, 9, 16, 16, device=dev, generator=torch.Generator(dev).manual_seed(1 + rank)).requires_grad_(True)
        out_grad = torch.randn(windows_rank, 16, 16, device=dev)
        params = []
        what = (f"cfg5: per-window left fold of 9 log-matrices (16x16) (logmatmulexp), float32, fwd + bwd, "
                f"{batch} samples x {CFG5_SITES} windows per GPU")

        def fwd_bwd():
            x.grad = None
            logmatmulexp_fold(x).backward(out_grad)

    reducer = (ddp.FlatGradAllReducer(params, skip_single_rank=not force_reduce, algorithm=args.allreduce)
               if (params and (world > 1 or force_reduce)) else None)
    direct = reducer is not None and getattr(reducer, "_direct", None) is not None

    def fwd_bwd_reduce():
        fwd_bwd()
        reducer()

    gsteps = 1
    if args.graph:   # the largest divisor of --steps that is <= --graph-steps: exactly K timed steps whatever K is
        gsteps = max(d for d in range(1, max(1, args.graph_steps) + 1) if args.steps % d == 0)

    def several(fn):
        def body():
            for _ in range(gsteps):
                fn()
        return body

    # Over RCCL the default is ONE graph per step with the collective inside (one host launch per step; the eager
    # collective made the step host-bound: 51.7 vs 42.3 us on one rank).  It is attempted only when the child-process
    # probe succeeded on every rank; if the capture then still fails, the run stops with a non-zero exit code instead
    # of limping on: after a failed capture of a collective, later collectives of this process fail.
    graph, reduce_in_graph = None, False
    if args.graph:
        # (the direct collective is a plain kernel launch: capturable without a probe)
        want_reduce = reducer is not None and (direct or (probe_ok is not None and ddp.all_ranks_agree(bool(probe_ok), dev)))
        if reducer is not None and args.graph_allreduce == "1" and not want_reduce:
            raise SystemExit("--graph-allreduce 1: the all-reduce capture probe failed (or the backend is not RCCL)")
        if want_reduce:
            fwd_bwd_reduce()
            torch.cuda.synchronize(dev)
            want = [p.grad.detach().float().clone() for p in params]
            graph = safe_capture(several(fwd_bwd_reduce), dev, warm=3)
            if not ddp.all_ranks_agree(graph is not None, dev):
                log("capturing the step with the all-reduce inside failed although the probe passed; no safe fallback "
                    "inside this process - rerun with --graph-allreduce 0")
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(3)
            graph.replay()
            torch.cuda.synchronize(dev)
            same = all(torch.allclose(p.grad.float(), w, rtol=2e-2, atol=1e-6 + 2e-2 * float(w.abs().max()))
                       for p, w in zip(params, want))
            if not ddp.all_ranks_agree(same, dev):
                log("the captured step with the all-reduce inside disagrees with the eager step")
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(4)
            reduce_in_graph = True
        else:
            if reducer is not None:
                gsteps = 1   # eager collective after every step: one step per graph
            graph = safe_capture(several(fwd_bwd), dev, warm=3)

    if graph is None:
        gsteps = 1

    def step():   # `gsteps` steps of the workload
        if graph is not None:
            graph.replay()
        else:
            fwd_bwd()
        if reducer is not None and not reduce_in_graph:
            reducer()

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(-(-args.warmup // gsteps)):   # at least W warm-up steps (rounded up to whole graph launches)
        step()
    kernel_used = dctn_amd.last_kernel()
    block_s = []
    for _ in range(BLOCKS):
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps // gsteps):
            step()
        torch.cuda.synchronize(dev)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t)
        block_s.append(elapsed)
    elapsed = statistics.median(block_s)

    # what the collective costs, alone: the same message, replayed from a graph when the step's collective is (device
    # time of back-to-back dependent all-reduces), else launched eagerly as in the step
    allreduce_us = step_without_allreduce_us = allreduce_bytes = other_us = None
    if reducer is not None:
        flat = getattr(reducer, "_flat", None)
        msg = flat if flat is not None else reducer.bucket
        allreduce_bytes = int(msg.numel() * msg.element_size())
        buf = torch.zeros_like(msg)

        def one_reduce():
            reducer._reduce(buf)

        if reduce_in_graph:
            def ten():
                for _ in range(10):
                    one_reduce()
            allreduce_us = device_time(ten, dev, 20) / 10 * 1e6
        else:
            allreduce_us = device_time(one_reduce, dev, 200, graph=False) * 1e6
        g2 = safe_capture(several(fwd_bwd), dev, warm=1) if args.graph else None   # same steps per launch as the step itself
        if g2 is not None:
            step_without_allreduce_us = device_time(g2.replay, dev, max(1, args.steps // gsteps), graph=False) / gsteps * 1e6
        else:
            step_without_allreduce_us = device_time(fwd_bwd, dev, args.steps, graph=False) * 1e6
        # the same message through the OTHER algorithm (eager launches; the direct one replayed from a graph too)
        other_us = None
        try:
            if direct and args.time_other_allreduce:
                other = ddp.FlatGradAllReducer(params, skip_single_rank=not force_reduce, algorithm="rccl")
                other_us = device_time(lambda: other._reduce(buf), dev, 100, graph=False) * 1e6
            elif world > 1 and msg.is_cuda and args.time_other_allreduce:
                # (DirectAllReducer's constructor is rank-symmetric: it raises on every rank or on none, so the ranks
                #  cannot part ways between its collectives and the MAX all-reduce below)
                dr = ddp.DirectAllReducer(msg.numel(), msg.dtype, dev, average=True)

                def ten_direct():
                    for _ in range(10):
                        dr(buf)
                other_us = device_time(ten_direct, dev, 20) / 10 * 1e6
                if not ddp.all_ranks_agree(dr.status() == 0, dev):
                    other_us = None
        except Exception as e:   # noqa: BLE001
            log(f"timing the other all-reduce algorithm failed ({type(e).__name__}: {e})")
        if direct and not ddp.all_ranks_agree(reducer._direct.status() == 0, dev):
            raise SystemExit("the step's direct all-reduce timed out waiting for a peer: the timed steps are invalid")
        if world > 1:   # the slowest rank's figures
            t = torch.tensor([allreduce_us, step_without_allreduce_us, other_us if other_us is not None else -1.0],
                             dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            allreduce_us, step_without_allreduce_us = float(t[0]), float(t[1])
            other_us = float(t[2]) if float(t[2]) >= 0 else None

    # one step per graph launch beside the default several: what a training loop that feeds fresh data every step gets
    # (GraphedTrainStep replays one iteration per launch); the difference is the command processor's per-launch work
    one_step_us = None
    if graph is not None and gsteps > 1 and (reducer is None or reduce_in_graph):
        g1 = safe_capture(fwd_bwd_reduce if reduce_in_graph else fwd_bwd, dev, warm=1)
        if g1 is not None:
            n1 = max(20, args.steps)
            for _ in range(5):
                g1.replay()
            barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n1):
                g1.replay()
            torch.cuda.synchronize(dev)
            one_step_us = (time.perf_counter() - t0) / n1 * 1e6
            del g1

    windows_step = windows_rank * world
    short = (f"{args.workload} EPS{list(specs)}+linear {DTYPE_NAME[dtype]} B{batch}/GPU".replace(", ", ",")
             if eps_model else what.split(",")[0][:60])   # (the full description: run.workload)
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
        "dtype": DTYPE_NAME[dtype],
        "data": "synthetic",
        # `config` stays SHORT (< 850 characters, asserted by tests/test_bench_host.py): the workload, then - at N = 1 -
        # three scalars per side configuration.  How the run went is in `run`, prose in `notes`.
        "config": {"workload": short, "parallelism": f"dp{world}"},
        "run": {
            "workload": what,
            "per_gpu_batch": batch,
            "windows_per_step": windows_step,
            "hip_graph": graph is not None,
            "steps_per_graph_launch": gsteps,
            "us_per_step_at_one_step_per_graph_launch": one_step_us,
            "blocks_ms": [b * 1e3 for b in block_s],
            "last_kernel": kernel_used,
        },
        "notes": {"timing": f"median of {BLOCKS} blocks of {args.steps} steps (each block: barrier + synchronize on both sides, max over ranks)"},
    }
    if reducer is not None or world > 1:
        line["run"].update({
            "allreduce_in_graph": reduce_in_graph,
            "allreduce_capture_probe": probe_ok,
            "allreduce": args.allreduce if reducer is not None else None,
            "allreduce_us": allreduce_us,
            "allreduce_us_direct": (allreduce_us if direct else other_us) if reducer is not None else None,
            "allreduce_us_rccl": (other_us if direct else allreduce_us) if reducer is not None else None,
            "allreduce_bytes": allreduce_bytes,
            "step_without_allreduce_us": step_without_allreduce_us,
            "rccl_env": rccl_env,
            "grad_allreduce": ("none: the workload has no parameters (windows sharded, no data-path collective)" if not params else None) if reducer is None else (
                "in place on the backward's flat gradient buffer (1 launch)" if getattr(reducer, "_flat_key", None)
                else "gather -> all_reduce -> scatter (3 launches)"),
        })
    if rank == 0:
        if eps_model:
            log(f"headline: {line['ms_per_step']*1e3:.1f} us/step; timing its kernels")
            line["roofline"] = headline_roofline(model, x, specs, args.steps, ms_per_step=line["ms_per_step"])
            if world == 1 and not args.no_cpu_baseline:
                log("cpu baseline of the headline workload")
                line["cpu_baseline"] = cpu_baseline_eps_model(specs, image_size, q0)
                line["notes"]["cpu_baseline"] = ("cpu_baseline runs the float32 oracle at batch 128 on the host cores; the GPU step is "
                                                 f"{DTYPE_NAME[dtype]} at batch {batch}")
        else:
            # ConvSBS / logmatmulexp as the timed workload: roofline and CPU baseline of the same configuration, measured
            # at its BASELINE size in a child process on this rank's GPU (the timed region is over)
            del x, out_grad
            torch.cuda.empty_cache()
            log(f"{args.workload}: {line['ms_per_step']*1e3:.1f} us/step; roofline / cpu baseline of the configuration")
            entry = run_extra_in_child(args.workload, args.no_cpu_baseline or world > 1, device_index=dev.index)
            if "roofline" in entry:
                line["roofline"] = entry["roofline"]
            if "cpu_baseline" in entry:
                line["cpu_baseline"] = entry["cpu_baseline"]
            line["run"]["single_gpu_entry"] = {k: entry.get(k) for k in ("ms_per_step", "value", "windows_per_step", "error") if k in entry}
        if world == 1 and args.configs != "none" and eps_model:
            names = EXTRA_CONFIGS if args.configs == "all" else tuple(n for n in args.configs.split(",") if n)
            entries = []
            for name in names:
                # one child process per configuration: a crash or an out-of-memory kill in a side configuration must not
                # take the headline number with it (the child only runs --skip-headline, it never re-executes this process)
                t0 = time.perf_counter()
                log(f"config {name}")
                entries.append(run_extra_in_child(name, args.no_cpu_baseline))
                entries[-1]["bench_seconds"] = round(time.perf_counter() - t0, 1)
            line["configs"] = entries
            # three scalars per BASELINE / SURVEY 8(d) configuration right behind the workload (what a record that keeps
            # ~900 characters of `config` still holds), and ALL of them once more as the line's LAST key, so that the
            # tail of stdout carries every configuration too
            line["config"] = compact_config(line["config"], names, entries)
            line["notes"]["side"] = ("config.side_<cfg>_ms: ms per fwd+bwd step; _frac: dominant call's fraction of its roofline; _cpu_wps: CPU "
                                     "oracle windows/s.  side_summary.<cfg> = [ms per step, windows/s, frac, bound, step_frac, "
                                     "counter traffic / algorithmic bytes, cpu windows/s]")
            line["side_summary"] = {name: side_summary_row(e) for name, e in zip(names, entries)}
        print(json.dumps(line), flush=True)
    barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

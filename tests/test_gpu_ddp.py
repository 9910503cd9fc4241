"""Two data-parallel ranks sharing ONE MI355X (gloo collectives on device tensors): the complete
training iteration — forward + fused backward replayed from a HIP graph, the in-place all-reduce of
the backward's flat gradient buffer, the optimizer graph — must leave both ranks with the parameters
a single process gets from the concatenated batch.  (RCCL needs one GPU per rank; on the one-GPU test
box the same code path runs over gloo.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(dtype, dev):
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd

    torch.manual_seed(5)
    return EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, dev, dtype, image_size=12)


def _data(dtype, dev):
    g = torch.Generator().manual_seed(17)
    u = torch.rand(1, 32, 12, 12, generator=g)
    x = torch.stack((torch.sin(u * torch.pi / 2) ** 2, torch.cos(u * torch.pi / 2) ** 2), dim=-1).to(dtype).to(dev)
    y = torch.randint(0, 10, (32,), generator=g).to(dev)
    return x, y


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist

    from dctn_amd import ddp
    from dctn_amd.training import FlatSGD, GraphedTrainStep, fused_cross_entropy

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ddp.init_from_env("gloo")
    dtype = torch.bfloat16
    model = _make(dtype, dev)
    ddp.broadcast_parameters(model.parameters())
    x, y = _data(dtype, dev)
    xs, ys = ddp.shard_batch(x, rank, world), y[rank * 16 : rank * 16 + 16]
    opt = FlatSGD(list(model.epses) + [model.linear.weight], [model.linear.bias], lr=0.05, momentum=0.9, l2=1e-3)
    red = ddp.FlatGradAllReducer(model.parameters(), average=True)
    step = GraphedTrainStep(model, xs, ys, fused_cross_entropy, opt, reducer=red, warmup=1)
    for _ in range(3):
        step(xs, ys)
    torch.cuda.synchronize(dev)
    assert step.g_opt is not None and red._flat_key is not None       # split graphs, in-place all-reduce
    # empirical-std initialisation on shards of the dataset: one global statistic, identical cores
    from dctn_amd.epses_composition import make_epses_composition_unit_empirical_output_std

    torch.manual_seed(100 + rank)   # different seeds on purpose: the random core must be rank 0's
    x32 = x.float()
    cores = make_epses_composition_unit_empirical_output_std(((2, 3), (2, 4)), ddp.shard_batch(x32, rank, world), dev,
                                                             torch.float32, batch_size=8)
    # by value (numpy): a tensor in the queue is a shared-memory handle the parent must fetch from this process
    q.put((rank, [p.detach().float().cpu().numpy() for p in model.parameters()] + [c.float().cpu().numpy() for c in cores]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_single_process_iteration():
    from dctn_amd.training import FlatSGD, GraphedTrainStep, fused_cross_entropy

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: [torch.from_numpy(a) for a in arrs] for r, arrs in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for a, b in zip(got[0], got[1]):
        assert torch.equal(a, b)
    # the two empirically initialised cores: unit output std over the WHOLE dataset, layer by layer
    from dctn_amd.eps import eps, transform_in_slices

    xin = _data(torch.bfloat16, torch.device("cuda", 0))[0].float()   # what the workers sharded
    for core in got[0][-2:]:
        out = transform_in_slices(core.to(xin.device), xin, 32)
        assert abs(float(out.std(unbiased=False)) - 1.0) < 1e-4
        xin = out
    got = {r: v[:-2] for r, v in got.items()}
    # single process, whole batch: 1 warm-up + 3 iterations, same optimizer
    dev = torch.device("cuda", 0)
    model = _make(torch.bfloat16, dev)
    x, y = _data(torch.bfloat16, dev)
    opt = FlatSGD(list(model.epses) + [model.linear.weight], [model.linear.bias], lr=0.05, momentum=0.9, l2=1e-3)
    step = GraphedTrainStep(model, x, y, fused_cross_entropy, opt, warmup=1)
    for _ in range(3):
        step(x, y)
    for a, p in zip(got[0], model.parameters()):
        ref = p.detach().float().cpu()
        assert float((a - ref).abs().max()) <= 3e-2 * float(ref.abs().max()), "ranks vs single process"


def _rccl_single_rank_worker(port, q):
    """One rank on the real backend (RCCL): the collective is issued although there is nothing to exchange, and the
    whole iteration — forward, fused backward, all-reduce, optimizer — is ONE captured graph."""
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from dctn_amd import ddp
    from dctn_amd.training import FlatSGD, GraphedTrainStep, fused_cross_entropy

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ddp.init_from_env("nccl", single_rank_group=True)
    model = _make(torch.bfloat16, dev)
    x, y = _data(torch.bfloat16, dev)
    opt = FlatSGD(list(model.epses) + [model.linear.weight], [model.linear.bias], lr=0.05, momentum=0.9, l2=1e-3)
    red = ddp.FlatGradAllReducer(model.parameters(), average=True, skip_single_rank=False)
    # graph_allreduce left at its default: a child-process probe decides, and over RCCL it must say yes
    step = GraphedTrainStep(model, x, y, fused_cross_entropy, opt, reducer=red, warmup=1)
    for _ in range(3):
        step(x, y)
    torch.cuda.synchronize(dev)
    q.put((step.allreduce_in_graph, step.g_opt is None, [p.detach().float().cpu().numpy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_single_rank_whole_iteration_in_one_graph():
    from dctn_amd.training import FlatSGD, GraphedTrainStep, fused_cross_entropy

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_rccl_single_rank_worker, args=(_free_port(), q))
    proc.start()
    in_graph, one_graph, params = q.get(timeout=300)
    proc.join(120)
    assert proc.exitcode == 0
    assert in_graph and one_graph
    dev = torch.device("cuda", 0)
    model = _make(torch.bfloat16, dev)
    x, y = _data(torch.bfloat16, dev)
    opt = FlatSGD(list(model.epses) + [model.linear.weight], [model.linear.bias], lr=0.05, momentum=0.9, l2=1e-3)
    step = GraphedTrainStep(model, x, y, fused_cross_entropy, opt, warmup=1)
    for _ in range(3):
        step(x, y)
    for a, p in zip(params, model.parameters()):
        assert torch.equal(torch.from_numpy(a), p.detach().float().cpu())   # averaging over one rank changes nothing


# ------------------------------------------------------------------ the ConvSBS classifier and the logmatmulexp fold
def _sbs_data(dev):
    g = torch.Generator().manual_seed(23)
    return torch.rand(1, 8, 9, 9, 2, generator=g).to(dev), torch.randint(0, 10, (8,), generator=g).to(dev)


def _sbs_model(dev):
    from tests.sbs_classifier import ConvSBSClassifier

    torch.manual_seed(5)
    m = ConvSBSClassifier(bond=4).to(dev)
    m.scales = [2.0, 3.0, 4.0]   # fixed (not calibrated per rank: both ranks and the single process must agree)
    return m


def _sbs_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist

    from dctn_amd import ddp
    from dctn_amd.logmatmulexp import logmatmulexp_fold

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ddp.init_from_env("gloo")
    model = _sbs_model(dev)
    if rank != 0:                              # deliberately different initial parameters: rank 0's are broadcast
        with torch.no_grad():
            for p in model.parameters():
                p.add_(torch.randn_like(p))
    ddp.broadcast_parameters(model.parameters())
    x, y = _sbs_data(dev)
    xs, ys = ddp.shard_batch(x, rank, world), y[rank * 4 : rank * 4 + 4]
    red = ddp.FlatGradAllReducer(model.parameters(), average=False)
    # eager step and a HIP-graph replay of forward + backward, each followed by the all-reduce of the gradients
    torch.nn.functional.cross_entropy(model(xs), ys, reduction="sum").backward()
    red()
    eager = [p.grad.detach().clone() for p in model.parameters()]
    # the gradients of ONE string sit back to back (the backward's flat buffer): a per-string reducer all-reduces in place
    s0 = model.layers[0].strings[0]
    red0 = ddp.FlatGradAllReducer(s0.parameters(), average=False)
    assert red0._contiguous_flat([p.grad for p in s0.parameters()]) is not None

    def fwd_bwd():
        for p in model.parameters():
            p.grad = None
        torch.nn.functional.cross_entropy(model(xs), ys, reduction="sum").backward()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fwd_bwd()
    g.replay()
    red()
    torch.cuda.synchronize(dev)
    for p, e in zip(model.parameters(), eager):
        assert torch.allclose(p.grad, e, rtol=1e-5, atol=1e-6 * float(e.abs().max()))
    # the logmatmulexp fold shards by windows with nothing to exchange: each rank folds its half
    gm = torch.Generator().manual_seed(3)
    mats = torch.randn(64, 9, 16, 16, generator=gm).to(dev)
    mine = logmatmulexp_fold(mats[rank * 32 : rank * 32 + 32])
    q.put((rank, [e.cpu().numpy() for e in eager], mine.cpu().numpy()))   # by value: the worker may exit before the parent reads
    dist.barrier()
    dist.destroy_process_group()


def test_convsbs_classifier_and_fold_two_ranks_on_one_gpu():
    """BASELINE configs[3] / [4] data-parallel: the reference's three-layer ConvSBS classifier (mnist.py:169-284) on two
    ranks sharing one MI355X (gloo): broadcast parameters, per-rank shard, all-reduced gradients = the single-process
    gradients of the whole batch; the window-sharded logmatmulexp fold = the single-process fold."""
    from dctn_amd.logmatmulexp import logmatmulexp_fold

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sbs_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (g, f) for r, g, f in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    dev = torch.device("cuda", 0)
    model = _sbs_model(dev)
    x, y = _sbs_data(dev)
    torch.nn.functional.cross_entropy(model(x), y, reduction="sum").backward()
    for a, b, p in zip(got[0][0], got[1][0], model.parameters()):
        a, b = torch.from_numpy(a), torch.from_numpy(b)
        assert torch.equal(a, b)
        ref = p.grad.cpu()
        assert torch.allclose(a, ref, rtol=2e-4, atol=2e-5 * float(ref.abs().max())), "ranks vs single process"
    gm = torch.Generator().manual_seed(3)
    mats = torch.randn(64, 9, 16, 16, generator=gm).to(dev)
    whole = logmatmulexp_fold(mats).cpu()
    assert torch.allclose(torch.cat([torch.from_numpy(got[0][1]), torch.from_numpy(got[1][1])]), whole, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("algorithm", ["rccl", "direct"])
def test_bench_gpus_2_starts_itself_without_world_size(algorithm):
    """`python bench.py --gpus 2` started the way the driver starts `--gpus 1` (one process, no WORLD_SIZE): the parent
    starts torch.distributed.run itself, before any GPU call of its own, and relays rank 0's one JSON line.  Rehearsed
    here with both ranks on the one GPU of the test box over gloo (RCCL needs a GPU per rank)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["DCTN_BENCH_ONE_DEVICE"] = "1"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "4",
                          "--warmup", "2", "--batch", "64", "--configs", "none", "--no-cpu-baseline", "--allreduce", algorithm,
                          "--time-other-allreduce", "1"],
                         cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-1000:]
    line = json.loads(lines[0])
    run = line["run"]   # how the run went (`config` holds the workload and the side configurations' scalars only)
    assert line["n_gpus"] == 2 and line["steps"] == 4 and run["windows_per_step"] == 2 * 64 * 676
    assert line["config"]["parallelism"] == "dp2" and "B64/GPU" in line["config"]["workload"]
    assert run["allreduce_bytes"] and line["value"] > 0 and run["allreduce"] == algorithm
    # both algorithms are timed on the step's message whichever one the step uses (`--backend gloo` stands in for RCCL here)
    assert run["allreduce_us_direct"] > 0 and run["allreduce_us_rccl"] > 0
    if algorithm == "direct":
        assert run["allreduce_in_graph"] is True   # a plain kernel launch: captured without a probe


# ------------------------------------------------------------------ the direct (one-shot) all-reduce
def _direct_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from dctn_amd import ddp

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ddp.init_from_env("gloo")
    out = {"rank": rank}
    try:
        import dctn_amd

        # ONE reducer (one IPC-exported block per rank) for every message: creating and destroying a block per case made
        # `hipIpcGetMemHandle` / the peers' mappings fail now and then on a cold box
        red = ddp.DirectAllReducer(1_900_000, torch.float32, dev, average=True)
        for dtype, n in ((torch.float32, 29098), (torch.bfloat16, 29099), (torch.float64, 1000), (torch.float32, 1_900_000)):
            tol = {torch.float32: 1e-6, torch.bfloat16: 8e-3, torch.float64: 1e-14}[dtype]
            results = {}
            # first the one-shot form, then - on the same inputs - the two-shot form (a chunk per rank, then the owners'
            # chunks), which must give bitwise the same values
            for form in ("one_shot", "two_shot"):
                g = torch.Generator().manual_seed(100 + rank)
                worst, mism, got = 0.0, [], []
                for step in range(5):   # (five steps: both staging buffers, the step counter, flag lines reused)
                    mine = torch.randn(n, generator=g).to(dtype)
                    both = [torch.empty(n, dtype=dtype) for _ in range(world)]
                    dist.all_gather(both, mine)
                    want = (sum(b.double() for b in both) / world)
                    buf = mine.to(dev)
                    red(buf, form=form)
                    torch.cuda.synchronize(dev)
                    assert dctn_amd.last_kernel() == ("allreduce_direct" if form == "one_shot" else "allreduce_direct_two_shot")
                    res = buf.cpu()
                    got.append(res)
                    worst = max(worst, float((res.double() - want).abs().max() / want.abs().max()) / tol)
                    # every rank holds bitwise the same result (same numbers added in the same order); recorded, not raised
                    # here: the ranks must stay in step for the collectives
                    mineb = res.view(torch.uint8)
                    allb = [torch.empty_like(mineb) for _ in range(world)]
                    dist.all_gather(allb, mineb)
                    if not all(torch.equal(allb[0], b) for b in allb):
                        mism.append(step)
                results[form] = got
                out[f"{dtype}_{n}_{form}"] = worst
                out[f"status_{dtype}_{n}_{form}"] = red.status()
                if mism:
                    out.setdefault("ranks_differ", []).append((str(dtype), n, form, mism))
            diff = [k for k, (a, b) in enumerate(zip(results["one_shot"], results["two_shot"])) if not torch.equal(a, b)]
            if diff:
                out.setdefault("two_shot_mismatch", []).append((str(dtype), n, diff))
        # replayed from a HIP graph: the step counter lives in device memory
        n = 29098
        static = torch.zeros(n, device=dev)
        side = torch.cuda.Stream(dev)
        with torch.cuda.stream(side):
            static.fill_(float(rank + 1))
            red(static)
        torch.cuda.synchronize(dev)
        dist.barrier()
        graph = torch.cuda.CUDAGraph()
        static.fill_(float(rank + 1))
        torch.cuda.synchronize(dev)
        with torch.cuda.graph(graph, stream=torch.cuda.Stream(dev), capture_error_mode="thread_local"):
            red(static)
        ok = True
        for k in range(4):
            static.fill_(float(rank + 1 + k))
            torch.cuda.synchronize(dev)
            dist.barrier()
            graph.replay()
            torch.cuda.synchronize(dev)
            wantv = sum(r + 1 + k for r in range(world)) / world
            ok = ok and bool((static == wantv).all())
        out["graph"] = ok
        out["status_graph"] = red.status()
        red.close()
        # through FlatGradAllReducer(algorithm="direct"): the same mean as the process group's all-reduce
        params = [torch.nn.Parameter(torch.randn(5, 7, device=dev)), torch.nn.Parameter(torch.randn(11, device=dev))]
        for p in params:
            p.grad = torch.full_like(p, float(rank + 1))
        fr = ddp.FlatGradAllReducer(params, average=True, algorithm="direct")
        fr()
        torch.cuda.synchronize(dev)
        out["flat"] = all(bool((p.grad == (world + 1) / 2).all()) for p in params)
    except Exception as e:   # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_direct_allreduce_two_ranks_on_one_gpu():
    """`dctn_ar_*` / `ddp.DirectAllReducer`: every rank's uncached block is mapped by its peer through an IPC handle; one
    kernel per step and rank copies, publishes its step number, waits for the peer's, and sums both staging buffers in
    rank order - here with two ranks sharing the one GPU of the test box (float32 / bfloat16 / float64, a 58 KB and a
    7.6 MB message, five steps each, a HIP-graph replay, and through `FlatGradAllReducer(algorithm="direct")`).  Every
    message also goes through the TWO-SHOT form (a chunk per rank, then the owners' chunks): bitwise the same values."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_direct_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for o in outs:
        assert "error" not in o, o
        statuses = {k: v for k, v in o.items() if k.startswith("status")}
        assert all(v == 0 for v in statuses.values()), f"a wait timed out (the two ranks' kernels did not overlap?): {statuses}"
        assert "ranks_differ" not in o, o["ranks_differ"]             # every rank holds bitwise the same result
        assert "two_shot_mismatch" not in o, o["two_shot_mismatch"]   # bitwise the one-shot values
        assert all(v <= 1.0 for k, v in o.items() if k.startswith("torch.")), o
        assert o["graph"] is True and o["flat"] is True, o


def _direct_worker_many(rank, world, port, q):
    """More than two ranks on the one GPU: the DEFAULT rule (`form="auto"`), element counts that neither the world size nor
    the 16-byte vectors divide, a buffer at an odd element offset (element accesses instead of 16-byte ones)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from dctn_amd import ddp

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ddp.init_from_env("gloo")
    out = {"rank": rank}
    try:
        import dctn_amd

        red = ddp.DirectAllReducer(400_000, torch.float32, dev, average=True)   # form="auto": the default rule
        cases = ((torch.float32, 150_003, 0), (torch.bfloat16, 300_005, 0), (torch.float64, 70_001, 0), (torch.float32, 1_001, 0),
                 (torch.float32, 150_003, 1), (torch.bfloat16, 300_005, 3))
        for dtype, n, off in cases:
            tol = {torch.float32: 1e-6, torch.bfloat16: 8e-3, torch.float64: 2e-15}[dtype]
            big = n * torch.empty((), dtype=dtype).element_size() >= (512 << 10)
            want_kernel = "allreduce_direct_two_shot" if (world >= 4 and big) else "allreduce_direct"
            g = torch.Generator().manual_seed(1000 + rank)
            for step in range(3):
                mine = torch.randn(n, generator=g).to(dtype)
                every = [torch.empty(n, dtype=dtype) for _ in range(world)]
                dist.all_gather(every, mine)
                want = sum(b.double() for b in every) / world
                res = {}
                for form in (None, "one_shot"):     # the default rule first, then the one-shot form on the same inputs
                    hold = torch.zeros(n + off, dtype=dtype, device=dev)
                    buf = hold[off:]                  # off > 0: not 16-byte aligned
                    buf.copy_(mine)
                    red(buf, form=form)
                    torch.cuda.synchronize(dev)
                    if form is None and dctn_amd.last_kernel() != want_kernel:
                        out.setdefault("wrong_form", []).append((str(dtype), n, dctn_amd.last_kernel()))
                    res[form] = buf.cpu()
                key = f"{dtype}_{n}_{off}"
                out[key] = max(out.get(key, 0.0), float((res[None].double() - want).abs().max() / want.abs().max()) / tol)
                if not torch.equal(res[None], res["one_shot"]):
                    out.setdefault("default_vs_one_shot", []).append((str(dtype), n, off, step))
                mineb = res[None].view(torch.uint8)
                allb = [torch.empty_like(mineb) for _ in range(world)]
                dist.all_gather(allb, mineb)
                if not all(torch.equal(allb[0], b) for b in allb):
                    out.setdefault("ranks_differ", []).append((str(dtype), n, off, step))
        out["status"] = red.status()
        red.close()
    except Exception as e:   # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 3])
def test_direct_allreduce_default_rule_more_ranks_on_one_gpu(world):
    """Four ranks time-sharing the one GPU take the DEFAULT rule into the two-shot form (world >= 4 and >= 512 KiB) with
    element counts that 4 does not divide (150 003 float32, 300 005 bfloat16, 70 001 float64): chunk remainders, the
    16-byte main part and its element tail, flag lines of more than two ranks - bitwise the one-shot values, identical on
    every rank, the float64 mean to float64 precision (three ranks: 1 / 3 is no power of two)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_direct_worker_many, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for o in outs:
        assert "error" not in o, o
        assert o["status"] == 0, f"a wait timed out: {o}"
        assert "wrong_form" not in o, o["wrong_form"]
        assert "default_vs_one_shot" not in o, o["default_vs_one_shot"]
        assert "ranks_differ" not in o, o["ranks_differ"]
        assert all(v <= 1.0 for k, v in o.items() if k.startswith("torch.")), o


def _late_rank_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time

    import torch.distributed as dist

    from dctn_amd import ddp

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ddp.init_from_env("gloo")
    out = {"rank": rank, "raised": None}
    try:
        params = [torch.nn.Parameter(torch.randn(300, device=dev))]
        params[0].grad = torch.full_like(params[0], float(rank + 1))
        fr = ddp.FlatGradAllReducer(params, average=True, algorithm="direct")   # check_every = 1: the default
        fr()                                   # in step: fine
        torch.cuda.synchronize(dev)
        out["first"] = bool((params[0].grad == (world + 1) / 2).all())
        dist.barrier()
        if rank == 1:
            time.sleep(3.5)                    # above the kernel's ~2 s wait bound: rank 0 gives up on this step
        try:
            fr()
            out["raised"] = False
        except RuntimeError as e:
            out["raised"] = str(e)
    except Exception as e:   # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_direct_allreduce_late_rank_raises_on_every_rank():
    """The direct kernel bounds its wait (~2 s) and then sums whatever is there: `FlatGradAllReducer(algorithm="direct")`
    reads the error word after the call (default `check_every=1`), agrees on it across the ranks and raises on ALL of
    them - the late rank included, whose own kernel saw nothing wrong."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_late_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=300) for _ in procs), key=lambda o: o["rank"])
    for p in procs:
        p.join(timeout=60)
    for o in outs:
        assert "error" not in o, o
        assert o["first"] is True
        assert isinstance(o["raised"], str) and "did not arrive" in o["raised"], o
    assert "gave up waiting for rank 1" in outs[0]["raised"] and "seen by another rank" in outs[1]["raised"]

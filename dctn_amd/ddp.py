"""Data-parallel training over the GPUs of one node: one process per GPU, `torch.distributed`
backend "nccl" (= RCCL over xGMI on ROCm), batch sharded per rank, ONE flat-bucket sum
all-reduce of all parameter gradients per step.

The reference has no distributed code at all; windows — hence samples — are
independent, so the only exchange the path needs is the gradient sum.  All parameters of the
models here total < 8 MB (BASELINE cfg2: 58 KB), so the all-reduce is latency-bound: a single
bucket, no per-layer overlap machinery.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


def init_from_env(backend: Optional[str] = None, single_rank_group: bool = False) -> tuple:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and creates the
    process group (for one rank only when `single_rank_group`: a rehearsal of the RCCL path on a
    one-GPU machine).  Returns (rank, local_rank, world_size)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def _probe_child_device(local_rank: Optional[int] = None) -> int:
    """GPU index the probe's child opens: the explicit argument, else the launcher's LOCAL_RANK, else - only when this
    process has ALREADY initialised the GPU (it then has a current device that means something) - that device, else 0.
    Never touches the GPU itself: the probe runs before `set_device` (bench.py), where `current_device()` is 0 on every
    rank and asking would create a HIP context on GPU 0 in every process."""
    if local_rank is not None:
        return int(local_rank)
    if os.environ.get("LOCAL_RANK") is not None:
        return int(os.environ["LOCAL_RANK"])
    if torch.cuda.is_initialized():
        return int(torch.cuda.current_device())
    return 0


def probe_allreduce_capture(timeout: float = 240.0, numel: int = 29098, dtype: torch.dtype = torch.bfloat16,
                            port_offset: int = 53, rank: Optional[int] = None, world_size: Optional[int] = None,
                            master_addr: Optional[str] = None, master_port: Optional[int] = None,
                            local_rank: Optional[int] = None) -> bool:
    """Whether an RCCL all-reduce can be captured into a HIP graph on this machine with this world size, found
    out in CHILD processes (`dctn_amd._probe_allreduce_capture`): every rank calls this at the same point, each
    starts one child on its own GPU, the children form a process group of their own on master_port + port_offset,
    capture + replay one all-reduce and report through their exit code.  A failed capture cannot be recovered
    from inside a process (later collectives fail), so the attempt is made where it is free.  Returns this
    rank's verdict; combine the ranks' verdicts with `all_ranks_agree` once the parent group exists.

    The children must form a group of the SAME size as the parent's: rank / world size come from the arguments, else
    from the parent's initialised process group, else from RANK / WORLD_SIZE; the rendezvous address from the arguments,
    else MASTER_ADDR / MASTER_PORT; the child's GPU from `local_rank`, else LOCAL_RANK, else the current device of
    an already initialised GPU runtime (`_probe_child_device`).  A world of more than one rank without a known rendezvous address (mp.spawn with an
    explicit init_method and no environment) cannot be probed: False, i.e. the collective stays outside the graph."""
    import subprocess
    import sys

    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else int(os.environ.get("RANK", "0"))
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else int(os.environ.get("WORLD_SIZE", "1"))
    if master_addr is None:
        master_addr = os.environ.get("MASTER_ADDR")
    if master_port is None and os.environ.get("MASTER_PORT"):
        master_port = int(os.environ["MASTER_PORT"])
    if world_size > 1 and (master_addr is None or master_port is None):
        return False   # nowhere for the children to meet: a world-1 probe would say nothing about this world
    if master_addr is None:
        master_addr = "127.0.0.1"
    if master_port is None:
        master_port = 29500
    local_rank = _probe_child_device(local_rank)
    # the children rendezvous among themselves: under torchrun the parent's environment says "use the agent's store"
    # (TORCHELASTIC_USE_AGENT_STORE), which on another port would wait for a server nobody starts
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_") and k != "TORCH_NCCL_ASYNC_ERROR_HANDLING"}
    env["MASTER_ADDR"] = str(master_addr)
    env["MASTER_PORT"] = str(int(master_port) + port_offset)
    env["RANK"] = str(int(rank))
    env["LOCAL_RANK"] = str(int(local_rank))
    env["WORLD_SIZE"] = str(int(world_size))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["DCTN_PROBE_NUMEL"] = str(int(numel))
    env["DCTN_PROBE_DTYPE"] = str(dtype).replace("torch.", "")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    try:
        child = subprocess.Popen([sys.executable, "-m", "dctn_amd._probe_allreduce_capture"], cwd=root, env=env,
                                 stdout=sys.stderr, stderr=sys.stderr)
    except OSError:
        return False
    try:
        return child.wait(timeout=timeout) == 0
    except subprocess.TimeoutExpired:
        child.kill()   # this exact child, by handle
        child.wait()
        return False


def all_ranks_agree(flag: bool, device: Optional[torch.device] = None) -> bool:
    """True iff `flag` is true on every rank (every rank must then take the same branch, or collectives stop
    pairing up)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def shard_batch(x: Tensor, rank: int, world: int, dim: int = 1) -> Tensor:
    """Even split of the batch dim (dim 1 of the (C,B,H,W,Q) layout); a remainder is dropped
    like the reference's DataLoader(drop_last=True) (dataset_loading.py:325)."""
    per = x.shape[dim] // world
    return x.narrow(dim, rank * per, per)


@torch.no_grad()
def broadcast_parameters(params: Iterable[Tensor], src: int = 0) -> None:
    """Identical initial parameters on every rank (the reference seeds one process)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for p in params:
        dist.broadcast(p.data, src=src)


class FlatGradAllReducer:
    """Sums (or averages) the gradients of `params` over all ranks through one flat buffer.

    The messages are tiny (BASELINE cfg2: 29 k values), so the step is latency-bound: the whole
    exchange is three launches - one `cat` into the bucket, one all-reduce (RCCL `AVG` when
    available), one multi-tensor copy back.  When all parameters share a dtype the bucket has that
    dtype (no conversion kernels); mixed dtypes go through an fp32 bucket.
    """

    def __init__(self, params: Iterable[Tensor], average: bool = True, skip_single_rank: bool = True,
                 algorithm: str = "rccl", check_every: int = 1):
        """algorithm: "rccl" (default: `dist.all_reduce`) or "direct" (`DirectAllReducer`: one kernel per step reading the
        peers' buffers over all xGMI links at once; ranks of one node, device tensors).

        check_every (direct only): the direct kernel bounds its wait for a late peer (~2 s) and then sets an error word
        instead of hanging the queue - the step's values are then WRONG.  Every `check_every`-th call (default: every
        call) `check()` reads that word, agrees on it across the ranks and raises `RuntimeError` on all of them, so a
        rank skew above the bound (a checkpoint on rank 0, a stalled data loader) stops the run instead of letting the
        replicas diverge.  The check synchronises the device; 0 disables it (the caller then calls `check()` itself:
        after every replay when the call sits inside a HIP graph, where this method does not run)."""
        self.params: List[Tensor] = [p for p in params if p.requires_grad]
        self.average = average
        self.algorithm = algorithm
        self.check_every = int(check_every)
        self._calls = 0
        self._direct: Optional["DirectAllReducer"] = None
        self.skip_single_rank = skip_single_rank   # False: issue the collective even for one rank (rehearsal)
        self._seen_grads: List[Optional[Tensor]] = []
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        numel = sum(p.numel() for p in self.params)
        ref = self.params[0]
        dtypes = {p.dtype for p in self.params}
        self.same_dtype = len(dtypes) == 1
        bucket_dtype = ref.dtype if self.same_dtype else (
            torch.float64 if torch.float64 in dtypes else torch.float32)
        self.bucket = torch.zeros(numel, dtype=bucket_dtype, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.bucket[off : off + p.numel()].view_as(p))
            off += p.numel()
        backend = dist.get_backend() if dist.is_initialized() else ""
        self.use_avg = average and backend == "nccl"
        if algorithm == "direct" and self.bucket.is_cuda and self.world > 1:
            # (created here, eagerly: it allocates, maps its peers' blocks and synchronises - none of which may happen
            # inside a HIP-graph capture of the step)
            self._direct = DirectAllReducer(numel, bucket_dtype, ref.device, average=average)

    def _reduce(self, buf: Tensor) -> None:
        """In-place sum / mean of ``buf`` over the ranks.  RCCL's AVG is one launch; if the backend or
        the dtype refuses it (raised synchronously at enqueue), fall back to SUM + divide for good."""
        if self._direct is not None and buf.is_cuda and buf.dtype == self._direct.dtype and buf.numel() <= self._direct.max_numel:
            self._direct(buf)
            self._calls += 1
            if self.check_every > 0 and self._calls % self.check_every == 0 and not torch.cuda.is_current_stream_capturing():
                self.check()
            return
        if self.use_avg:
            try:
                dist.all_reduce(buf, op=dist.ReduceOp.AVG)
                return
            except (RuntimeError, ValueError):
                self.use_avg = False
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        if self.average:
            buf.div_(self.world)

    def check(self) -> None:
        """Direct algorithm only: raises `RuntimeError` on EVERY rank when a wait of any rank's all-reduce kernel timed
        out since the block was created (the gradients of that step are invalid).  Synchronises the device and runs one
        small collective; a no-op for the RCCL path (RCCL waits)."""
        if self._direct is None:
            return
        code = self._direct.status()
        if not all_ranks_agree(code == 0, self.bucket.device):
            raise RuntimeError(
                "direct all-reduce: a rank did not arrive within the kernel's wait bound"
                + (f" (this rank gave up waiting for rank {code - 1})" if code > 0 else " (seen by another rank)")
                + "; the gradients of that step are invalid - restart from the last checkpoint, or use algorithm='rccl'")

    def _contiguous_flat(self, grads) -> Optional[Tensor]:
        """One 1-D tensor aliasing all gradients when they already sit back to back, in parameter
        order, in a single storage (the fused EPS + head backward allocates them like that)."""
        if not self.same_dtype or any(g is None or not g.is_contiguous() for g in grads):
            return None
        g0 = grads[0]
        esz, base, off = g0.element_size(), g0.data_ptr(), 0
        st = g0.untyped_storage()
        for g in grads:
            if g.untyped_storage().data_ptr() != st.data_ptr() or g.data_ptr() != base + off * esz:
                return None
            off += g.numel()
        key = (base, off)
        if getattr(self, "_flat_key", None) != key:
            self._flat = g0.new_empty(0).set_(st, g0.storage_offset(), (off,), (1,))
            self._flat_key = key
        return self._flat

    @torch.no_grad()
    def __call__(self) -> None:
        if self.world == 1 and self.skip_single_rank:
            return
        grads = [p.grad for p in self.params]
        # a replayed HIP graph leaves the very same gradient tensors in place: skip the layout checks
        if len(grads) == len(self._seen_grads) and all(a is b for a, b in zip(grads, self._seen_grads)):
            flat = self._flat
        else:
            flat = self._contiguous_flat(grads)
            self._seen_grads = grads if flat is not None else []
        if flat is not None:   # the backward already laid the gradients out as one bucket: one launch
            self._reduce(flat)
            return
        if self.same_dtype and all(g is not None for g in grads):
            torch.cat([g.reshape(-1) for g in grads], out=self.bucket)
        else:
            for g, v in zip(grads, self.views):
                if g is None:
                    v.zero_()
                else:
                    v.copy_(g)
        self._reduce(self.bucket)
        if self.same_dtype and all(g is not None for g in grads):
            torch._foreach_copy_(grads, self.views)
        else:
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    p.grad = v.to(p.dtype).clone()
                else:
                    p.grad.copy_(v)


class DirectAllReducer:
    """One-shot direct all-reduce of ONE flat device tensor over peer-mapped buffers (`dctn_ar_*`, SURVEY section 5: the
    messages are 58 KB .. 7.5 MB, latency-bound; on a fully connected xGMI node every rank reads its peers' buffers over all
    its links at once instead of paying the 2 (P - 1) hops of a ring).  The ranks must share a node (IPC handles are
    exchanged through the process group with `all_gather_object`; two ranks on one GPU work too).  `__call__(buf)` enqueues
    one kernel on the current stream, in place, capturable into a HIP graph; every rank gets bitwise the same mean (or sum).
    `status()` synchronises and returns 0, or r + 1 when a wait for rank r timed out (~2 s) - that step's values are then
    invalid and the caller should fall back to the RCCL path.  `form`: "auto" (two-shot - every rank reduces one chunk, then
    the chunks are copied from their owners - for world >= 4 and >= 512 KiB, else one-shot), "one_shot", "two_shot";
    both forms give bitwise the same values."""

    FORMS = {"auto": 0, "one_shot": 1, "two_shot": 2}

    def __init__(self, max_numel: int, dtype: torch.dtype, device: torch.device, average: bool = True, form: str = "auto"):
        import ctypes

        from . import _lib as L

        assert dist.is_initialized(), "DirectAllReducer needs a process group to exchange its IPC handles"
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.average, self.device, self.dtype = average, device, dtype
        self.form = self.FORMS[form]
        self._L = L
        lib = L.lib()
        nbytes = int(max_numel) * torch.empty((), dtype=dtype).element_size()
        state = ctypes.c_void_p()
        self._state = None
        # Rank-symmetric on failure: a rank whose create / export / connect fails still takes part in every collective
        # of this constructor (it contributes an empty handle), the ranks agree on the outcome after each phase, and
        # ALL of them raise together - a failing rank never leaves its peers blocked in all_gather_object / barrier.
        with torch.cuda.device(device):
            hb = int(lib.dctn_ar_handle_bytes())
            mine = ctypes.create_string_buffer(hb)
            rc = lib.dctn_ar_create(self.world, self.rank, nbytes, ctypes.byref(state))
            if rc == 0:
                self._state = state
                rc = lib.dctn_ar_export(state, mine)
            handles = [None] * self.world
            dist.all_gather_object(handles, (rc, bytes(mine.raw)))
            failed = [r for r, (code, _) in enumerate(handles) if code != 0]
            if failed:
                self.close()
                raise RuntimeError(f"direct all-reduce: create / export failed on rank(s) {failed} "
                                   f"({L.lib().dctn_strerror(handles[failed[0]][0]).decode()})")
            packed = ctypes.create_string_buffer(b"".join(h for _, h in handles), hb * self.world)
            rc = lib.dctn_ar_connect(state, packed)
            if not all_ranks_agree(rc == 0, device if dist.get_backend() == "nccl" else None):
                self.close()
                raise RuntimeError("direct all-reduce: mapping the peers' blocks failed on at least one rank"
                                   + (f" (this rank: {L.lib().dctn_strerror(rc).decode()})" if rc != 0 else ""))
        dist.barrier()   # every rank has mapped every block before the first kernel publishes into one
        self.max_numel = int(max_numel)
        self.max_bytes = nbytes

    def __call__(self, buf: Tensor, form: str = None) -> None:
        """In place.  Any of float32 / float64 / bfloat16 that fits the block (`max_numel` elements of the constructor's
        dtype); `form` overrides the constructor's for this call.  All ranks must make the same calls in the same order."""
        assert buf.is_cuda and buf.is_contiguous() and buf.numel() * buf.element_size() <= self.max_bytes
        L = self._L
        L.check(L.lib().dctn_ar_allreduce_algo(self._state, buf.data_ptr(), buf.numel(), L.dtype_code(buf), int(self.average),
                                                self.form if form is None else self.FORMS[form], L.stream_ptr(buf.device)),
                "direct all-reduce")

    def status(self) -> int:
        return int(self._L.lib().dctn_ar_status(self._state))

    def close(self) -> None:
        if getattr(self, "_state", None) is not None:
            self._L.lib().dctn_ar_destroy(self._state)
            self._state = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@torch.no_grad()
def all_reduce_scalar_sums(*values: Tensor) -> List[Tensor]:
    """Sum of evaluation statistics (summed CE, correct counts: evaluation.py:18-19) over ranks."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(values)
    packed = torch.stack([v.double().reshape(()) for v in values])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    return list(packed.unbind(0))


@torch.no_grad()
def global_biased_std_from_sums(count: int, sums: Tensor) -> Tensor:
    """The same statistic from ``(count, [sum, sum of squares])`` of this rank's shard (what `dctn_eps_fwd_stats`
    leaves): one all-reduce of three float64 numbers."""
    n, s1, s2 = all_reduce_scalar_sums(torch.tensor(float(count), device=sums.device), sums[0], sums[1])
    mean = s1 / n
    return (s2 / n - mean * mean).clamp_min(0).sqrt()


@torch.no_grad()
def global_biased_std(values: Tensor) -> Tensor:
    """Biased standard deviation of the union of every rank's ``values`` (the statistic behind the
    empirical-std initialisation, dctn/eps.py:163-181, when each rank holds a shard of the dataset):
    count, sum and sum of squares are accumulated in float64 and summed over the ranks."""
    v = values.double()
    n, s1, s2 = all_reduce_scalar_sums(torch.tensor(float(v.numel()), device=v.device), v.sum(), (v * v).sum())
    mean = s1 / n
    return (s2 / n - mean * mean).clamp_min(0).sqrt()

"""One training iteration, data-parallel.

The reference's loop (dctn/training.py:62-84) is: forward, ``loss_fn(output, y)``, a regulariser
term scaled by ``reg_coeff``, ``optimizer.zero_grad()``, backward of the sum, ``optimizer.step()``,
with callback hooks around it.  Here the same sequence is one function; between backward and the
optimizer step the parameter gradients are averaged over the ranks (``ddp.FlatGradAllReducer``: in
place on the fused backward's flat gradient buffer when the model provides one).

``train`` and the callbacks below it mirror the reference's loop and hook objects
(dctn/training.py:23-248: same names, arguments, ``st_x`` / ``st_it`` dictionaries and checkpoint file
names) for SURVEY 8(f) row f4, made safe for one process per GPU: gradients are averaged over the ranks
before the ``after_back`` hooks, a stop requested on any rank stops all of them in the same iteration,
and only rank 0 touches checkpoint files.
"""
from __future__ import annotations

import os
from collections import deque
from logging import getLogger
from typing import Any, Callable, Dict, Iterable, Iterator, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from . import ddp

StX = Dict[Any, Any]   # state that lives across iterations
StIt = Dict[Any, Any]  # state of one iteration


def train_step(
    model: torch.nn.Module,
    x: Tensor,
    y: Tensor,
    loss_fn: Callable[[Tensor, Tensor], Tensor],
    optimizer: torch.optim.Optimizer,
    reg_fn: Optional[Callable[[torch.nn.Module], Tensor]] = None,
    reg_coeff: float = 0.0,
    reducer: Optional[ddp.FlatGradAllReducer] = None,
) -> Dict[str, Tensor]:
    """``x``: this rank's shard (channels, batch, height, width, features); ``y``: its labels.
    Returns the detached ``output``, ``loss`` and ``reg_term`` of this rank."""
    model.train()
    output = model(x)
    loss = loss_fn(output if getattr(loss_fn, "accepts_low_precision", False) else output.float(), y)
    optimizer.zero_grad(set_to_none=True)
    if reg_fn is not None:
        reg_term = reg_fn(model)
        (loss + reg_term.float() * reg_coeff).backward()
    else:   # no fill / multiply / add launches for a term that is not there
        reg_term = output.new_zeros((), dtype=torch.float32)
        loss.backward()
    if reducer is not None:
        reducer()
    optimizer.step()
    return {"output": output.detach(), "loss": loss.detach(), "reg_term": reg_term.detach()}


class GraphedTrainStep:
    """The same iteration replayed from captured HIP graphs (the step of the small models is launch
    bound: ~40 kernels of a few microseconds each once loss, regulariser and optimizer are counted).

    Single process: one graph holds forward, loss, regulariser, backward and the optimizer step.
    Data parallel over RCCL (``graph_allreduce=None``, the default): the collective is captured into the one graph
    as well (one launch per iteration from the host) when a child-process probe (`ddp.probe_allreduce_capture`,
    every rank at the same point) shows that such a capture works on this machine - a capture that fails cannot be
    recovered from inside a process (later collectives fail), so it is tried where a failure is free; when the
    probe fails, or with another backend, or with ``graph_allreduce=False``: forward + backward are one graph,
    the gradient all-reduce runs eagerly on the same stream, the optimizer step is a second graph.
    ``graph_allreduce=True`` skips the probe and insists.
    Inputs are copied into static buffers, so
    every call must use the batch shape of the example; the optimizer must be capturable
    (``torch.optim.SGD``, or ``Adam(..., capturable=True)``).

    The dictionary a call returns holds the graph's STATIC output buffers (no copy kernels in the iteration): the
    next call overwrites them.  A caller that keeps ``loss`` / ``output`` across iterations must ``.clone()`` them
    (``train_step`` returns fresh tensors).
    """

    def __init__(self, model: torch.nn.Module, example_x: Tensor, example_y: Tensor,
                 loss_fn: Callable[[Tensor, Tensor], Tensor], optimizer: torch.optim.Optimizer,
                 reg_fn: Optional[Callable[[torch.nn.Module], Tensor]] = None, reg_coeff: float = 0.0,
                 reducer: Optional[ddp.FlatGradAllReducer] = None, warmup: int = 3,
                 graph_allreduce: Optional[bool] = None):
        self.model, self.optimizer, self.reducer = model, optimizer, reducer
        self.x, self.y = example_x.clone(), example_y.clone()
        dev = example_x.device
        reduces = reducer is not None and (reducer.world > 1 or not reducer.skip_single_rank)
        if reduces and graph_allreduce is None:
            graph_allreduce = False
            if dist.is_initialized() and dist.get_backend() == "nccl":
                some = next(p for p in model.parameters() if p.requires_grad)
                ok = ddp.probe_allreduce_capture(numel=sum(p.numel() for p in model.parameters() if p.requires_grad),
                                                 dtype=some.dtype, local_rank=dev.index)
                graph_allreduce = ddp.all_ranks_agree(ok, dev)
        graph_allreduce = bool(graph_allreduce) and reduces
        split = reduces

        def fwd_bwd():
            model.train()
            out = model(self.x)
            loss = loss_fn(out if getattr(loss_fn, "accepts_low_precision", False) else out.float(), self.y)
            optimizer.zero_grad(set_to_none=True)
            if reg_fn is not None:
                reg = reg_fn(model)
                total = loss + reg.float() * reg_coeff
            else:   # no fill / multiply / add nodes in the graph for a term that is not there
                reg, total = no_reg, loss
            # a ready-made "1" (created during the eager warm-up, never inside the capture): autograd's own root
            # gradient would be a fill node
            total.backward(unit_seed(dev, total.dtype))
            return out, loss, reg

        no_reg = torch.zeros((), dtype=torch.float32, device=dev)
        assert warmup >= 1, "capture needs at least one eager iteration first (lazy optimizer state, kernel attributes)"
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fwd_bwd()
                if split:
                    reducer()
                optimizer.step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.g_opt = None
        self.allreduce_in_graph = False
        if reduces and graph_allreduce:
            try:   # the whole iteration, collective included, as one graph
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self.out, self.loss, self.reg = fwd_bwd()
                    reducer()
                    optimizer.step()
                self.g_main, self.allreduce_in_graph, split = g, True, False
                return
            except Exception as e:
                # no fallback: after a failed capture later collectives of this process fail ("invalid argument")
                raise RuntimeError("capturing the all-reduce into the iteration's graph failed; rerun with "
                                   "graph_allreduce=False") from e
        self.g_main = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_main, capture_error_mode="thread_local"):
            self.out, self.loss, self.reg = fwd_bwd()
            if not split:
                optimizer.step()
        if split:
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, capture_error_mode="thread_local"):
                optimizer.step()

    def __call__(self, x: Tensor, y: Tensor) -> Dict[str, Tensor]:
        if x is not self.x:   # a data pipeline may fill the static buffers `step.x` / `step.y` itself and pass them
            self.x.copy_(x, non_blocking=True)
        if y is not self.y:
            self.y.copy_(y, non_blocking=True)
        self.g_main.replay()
        if self.g_opt is not None:
            self.reducer()
            self.g_opt.replay()
        return {"output": self.out, "loss": self.loss, "reg_term": self.reg}


# ------------------------------------------------------------------------------- fused iteration tail
from . import _lib as L  # noqa: E402


# Constant "1" tensors used as the gradient seed of a scalar loss (GraphedTrainStep), one per (device, dtype), kept alive
# for the life of the process so that their addresses are never reused: a backward that receives one of them knows
# its incoming gradient is exactly 1 without reading it.
_UNIT_SEED_TENSORS: Dict = {}
_UNIT_SEEDS = set()


def unit_seed(device: torch.device, dtype: torch.dtype) -> Tensor:
    key = (str(device), dtype)
    seed = _UNIT_SEED_TENSORS.get(key)
    if seed is None:
        seed = _UNIT_SEED_TENSORS[key] = torch.ones((), dtype=dtype, device=device)
        _UNIT_SEEDS.add(seed.data_ptr())
    return seed


class _FusedCrossEntropy(torch.autograd.Function):
    """``F.cross_entropy(logits, labels)`` (mean reduction) as HIP kernels instead of cast + log-softmax + nll and
    their three backward launches.  When the logits need a gradient the forward kernel also leaves
    ``(softmax - onehot) / B`` (`dctn_ce_loss_fwd_grad`); the backward returns it as is when the incoming gradient is
    a registered constant 1 (the seed GraphedTrainStep passes), otherwise `dctn_ce_loss_bwd` scales by the incoming
    scalar."""

    @staticmethod
    def forward(ctx, logits: Tensor, labels: Tensor) -> Tensor:
        dev = L.require_device(logits, labels)
        lg, lb = logits.contiguous(), labels.contiguous().long()
        assert lg.ndim == 2 and lb.shape == (lg.shape[0],)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        if ctx.needs_input_grad[0]:
            unit = torch.empty_like(lg)
            L.check(L.lib().dctn_ce_loss_fwd_grad(lg.data_ptr(), lb.data_ptr(), loss.data_ptr(), unit.data_ptr(),
                                                  lg.shape[0], lg.shape[1], L.dtype_code(lg), L.stream_ptr(dev)),
                    "cross-entropy forward")
            ctx.save_for_backward(lg, lb, unit)
        else:
            L.check(L.lib().dctn_ce_loss_fwd(lg.data_ptr(), lb.data_ptr(), loss.data_ptr(), lg.shape[0], lg.shape[1],
                                             L.dtype_code(lg), L.stream_ptr(dev)), "cross-entropy forward")
        return loss

    @staticmethod
    def backward(ctx, d_loss: Tensor):
        lg, lb, unit = ctx.saved_tensors
        if d_loss.data_ptr() in _UNIT_SEEDS:
            # inside a capture the saved tensor itself is handed on (one node less; the graph owns it).  Eagerly it is
            # cloned: AccumulateGrad may steal a returned gradient and accumulate into it in place, which would
            # corrupt the saved tensor for a second backward(retain_graph=True)
            return (unit if torch.cuda.is_current_stream_capturing() else unit.clone()), None
        dev = lg.device
        g = d_loss.to(torch.float32).contiguous()
        d_logits = torch.empty_like(lg)
        L.check(L.lib().dctn_ce_loss_bwd(lg.data_ptr(), lb.data_ptr(), g.data_ptr(), d_logits.data_ptr(), lg.shape[0],
                                         lg.shape[1], L.dtype_code(lg), L.stream_ptr(dev)), "cross-entropy backward")
        return d_logits, None


def fused_cross_entropy(logits: Tensor, labels: Tensor) -> Tensor:
    """Mean cross-entropy of (batch, classes) logits (float32 or bfloat16) as a float32 scalar, with
    `F.cross_entropy`'s defaults: rows labelled -100 (its ignore_index) add nothing, get a zero gradient and do not count
    in the mean.  Any other label outside [0, classes) makes the loss and that row's gradient NaN (torch raises there;
    the reference's loaders never produce one: dctn/dataset_loading.py:282-286)."""
    return L.on_device(_FusedCrossEntropy.apply, logits, labels)


fused_cross_entropy.accepts_low_precision = True   # train_step / GraphedTrainStep skip their float32 cast


class FlatSGD:
    """SGD with momentum plus the reference's L2 regulariser, as ONE kernel per step over one flat
    parameter buffer (`dctn_sgd_l2_step`).

    ``regularised``: the parameters whose squared Frobenius norms the regulariser sums (for
    EPSesPlusLinear.epswise_l2_regularizer: every core and ``linear.weight``,
    dctn/eps_plus_linear.py:149-153); ``others``: the rest (``linear.bias``).  Adding
    ``l2 * sum ||w||^2`` to the loss and letting autograd differentiate it is the same update as adding
    ``2 * l2 * w`` to the gradient, which is what the kernel does; ``reg_value()`` returns the term's
    value (before the update) for logging.  The parameters are moved into one buffer (their ``.data``
    become views of it, in the order regularised + others); when the gradients already sit back to back
    in that order (the fused EPS + head backward allocates them so) the step reads them in place,
    otherwise they are gathered first.  Semantics of torch.optim.SGD(momentum, dampening = 0).
    """

    def __init__(self, regularised, others=(), lr: float = 1e-3, momentum: float = 0.0, l2: float = 0.0):
        self.reg_params = [p for p in regularised]
        self.params = self.reg_params + [p for p in others]
        assert self.params and len({p.dtype for p in self.params}) == 1 and len({p.device for p in self.params}) == 1
        self.lr, self.momentum, self.l2 = float(lr), float(momentum), float(l2)
        ref = self.params[0]
        self.n = sum(p.numel() for p in self.params)
        self.n_reg = sum(p.numel() for p in self.reg_params)
        self.flat = torch.empty(self.n, dtype=ref.dtype, device=ref.device)
        off = 0
        with torch.no_grad():
            for p in self.params:
                view = self.flat[off : off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                off += p.numel()
        self.buf = torch.zeros(self.n, dtype=torch.float32, device=ref.device)
        # one partial sum of squares per workgroup of the kernel; added up only when the value is asked for
        self.sq_sum = torch.zeros(L.lib().dctn_sgd_l2_num_partials(self.n), dtype=torch.float32, device=ref.device)
        self.flat_grad = torch.zeros(self.n, dtype=ref.dtype, device=ref.device)
        self._steps = 0

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _grads(self) -> Tensor:
        grads = [p.grad for p in self.params]
        base, off, st = None, 0, None
        ok = all(g is not None and g.is_contiguous() for g in grads)
        if ok:
            g0 = grads[0]
            base, st = g0.data_ptr(), g0.untyped_storage()
            for g in grads:
                if g.untyped_storage().data_ptr() != st.data_ptr() or g.data_ptr() != base + off * g.element_size():
                    ok = False
                    break
                off += g.numel()
        if ok:
            return grads[0].new_empty(0).set_(st, grads[0].storage_offset(), (self.n,), (1,))
        torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(grads, self.params)],
                  out=self.flat_grad)
        return self.flat_grad

    @torch.no_grad()
    def step(self) -> None:
        g = self._grads()
        dev = self.flat.device
        L.check(L.lib().dctn_sgd_l2_step(self.flat.data_ptr(), g.data_ptr(), self.buf.data_ptr(), self.sq_sum.data_ptr(),
                                         self.n, self.n_reg, self.lr, self.momentum, self.l2,
                                         1 if self._steps == 0 else 0, L.dtype_code(self.flat), L.stream_ptr(dev)),
                "fused SGD step")
        self._steps += 1

    def reg_value(self) -> Tensor:
        """l2 * sum of squared Frobenius norms of the regularised parameters, as of the last step."""
        return self.sq_sum.sum() * self.l2


# ------------------------------------------------------------------------------------------------
# The reference's training loop and its hooks (dctn/training.py:14-248), one process per GPU
# ------------------------------------------------------------------------------------------------
def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _is_rank0() -> bool:
    return _world() == 1 or dist.get_rank() == 0


def batches_forever(dl: Iterable) -> Iterator[Any]:
    """Cycles through ``dl`` again and again (a new pass — a new shuffle — each time it runs out)."""
    while True:
        for batch in dl:
            yield batch


def train(dl, model, optimizer, dev, loss_fn, reg_fn, reg_coeff: float, at_iter_start, after_back,
          after_param_upd) -> Tuple[StX, StIt]:
    """The loop of dctn/training.py:23-84.  ``dl`` yields ``(x, y, indices)`` (this rank's shard under
    ``torch.distributed``); ``loss_fn(output, y)`` and ``reg_fn(st_x, st_it)`` return 0-dim tensors; the three
    hook lists hold callables ``f(st_x, st_it)`` run at the start of an iteration, after ``backward()`` (the
    gradients they see are already averaged over the ranks) and after ``optimizer.step()``.  A hook stops
    the loop by setting ``st_it["stop"]``; with several ranks the flag is OR-ed over them so that all
    leave in the same iteration.  Returns the two state dictionaries of the last iteration."""
    st_x: StX = dict(model=model.to(dev), optimizer=optimizer, loss_fn=loss_fn, reg_fn=reg_fn, reg_coeff=reg_coeff,
                     at_iter_start=list(at_iter_start), after_back=list(after_back),
                     after_param_upd=list(after_param_upd), dev=dev)
    world = _world()
    reducer = None
    if world > 1:
        # the reference is one process with one seed; here every rank drew its own random parameters, and averaged
        # gradients on different models would diverge silently: everybody starts from rank 0's model
        ddp.broadcast_parameters(list(st_x["model"].parameters()) + list(st_x["model"].buffers()))
        if hasattr(st_x["model"], "_refresh_p"):
            st_x["model"]._refresh_p()
        reducer = ddp.FlatGradAllReducer(st_x["model"].parameters(), average=True)
    st_it: StIt = {}

    def run_hooks(key: str) -> None:
        for hook in tuple(st_x[key]):   # a hook may remove itself
            hook(st_x, st_it)

    for count, (x, y, indices) in enumerate(batches_forever(dl)):
        st_it = dict(num_iters_done=count, x=x.to(dev), y=y.to(dev), indices=indices.to(dev), stop=False)
        run_hooks("at_iter_start")
        st_x["model"].train()
        st_it["output"] = st_x["model"](st_it["x"])
        st_it["loss"] = st_x["loss_fn"](st_it["output"], st_it["y"])
        st_it["reg_term"] = st_x["reg_fn"](st_x, st_it)
        st_x["optimizer"].zero_grad()
        (st_it["loss"] + st_it["reg_term"] * st_x["reg_coeff"]).backward()
        if reducer is not None:
            reducer()
        run_hooks("after_back")
        st_x["optimizer"].step()
        run_hooks("after_param_upd")
        if world > 1:
            flag = torch.tensor([1.0 if st_it["stop"] else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            st_it["stop"] = bool(flag.item() > 0)
        if st_it["stop"]:
            break
    return st_x, st_it


def every_n_iters_intervals(*intervals):
    """Decorator factory: ``intervals`` are ``(length, period)`` pairs laid end to end from iteration 0; the
    hook runs when ``num_iters_done`` is a multiple of the period of the interval it falls in.  The last
    length may be ``None`` (forever); otherwise "every iteration" continues after the last interval."""
    spans = list(intervals)
    if spans[-1][0] is not None:
        spans.append((None, 1))
    starts, begin = [], 0
    for length, _ in spans:
        starts.append(begin)
        begin = begin + length if length is not None else begin
    periods = [period for _, period in spans]

    def decorate(func: Callable[[StX, StIt], None]) -> Callable[[StX, StIt], None]:
        def gated(st_x: StX, st_it: StIt) -> None:
            n = st_it["num_iters_done"]
            period = periods[0]
            for start, candidate in zip(starts, periods):
                if n >= start:
                    period = candidate
            if n % period == 0:
                func(st_x, st_it)

        return gated

    return decorate


def _checkpoint_tag(st_it: StIt) -> str:
    """The metric part of the reference's checkpoint names (training.py:135-140,161-166)."""
    return (f"nitd={st_it['num_iters_done']:07}_tracc={st_it['train_acc']:.4f}_vacc={st_it['val_acc']:.4f}_"
            f"trmce={st_it['train_mean_ce']:.4f}_vmce={st_it['val_mean_ce']:.4f}")


class Checkpointer:
    """Writes ``model.state_dict()`` files into ``dir`` — on rank 0 only; the other ranks hold the same
    parameters and skip the file system."""

    def __init__(self, dir: str):
        self.dir = dir

    def save(self, st_x: StX, filename: str) -> None:
        if _is_rank0():
            torch.save(st_x["model"].state_dict(), os.path.join(self.dir, filename))

    def remove_file(self, filename: str) -> None:
        if _is_rank0():
            os.remove(os.path.join(self.dir, filename))


class LastModelsCheckpointer(Checkpointer):
    """One checkpoint per call, the newest ``n`` are kept."""

    def __init__(self, dir: str, n: int):
        super().__init__(dir)
        assert n >= 1
        self.n = n
        self.filenames: deque = deque()

    def __call__(self, st_x: StX, st_it: StIt) -> None:
        name = f"model_{_checkpoint_tag(st_it)}.pth"
        self.save(st_x, name)
        self.filenames.appendleft(name)
        while len(self.filenames) > self.n:
            self.remove_file(self.filenames.pop())


class BestModelCheckpointer(Checkpointer):
    """Keeps the single checkpoint with the best ``st_it[key]`` seen so far."""

    def __init__(self, dir: str, key: str, low_is_good: bool):
        super().__init__(dir)
        self.key, self.low_is_good = key, low_is_good
        self.best_value = float("inf") if low_is_good else float("-inf")
        self.filename: Optional[str] = None

    def __call__(self, st_x: StX, st_it: StIt) -> None:
        value = st_it[self.key]
        better = value < self.best_value if self.low_is_good else value > self.best_value
        if not better:
            return
        name = f"model_best_{self.key}_{_checkpoint_tag(st_it)}.pth"
        self.save(st_x, name)
        self.best_value = value
        if self.filename is not None:
            self.remove_file(self.filename)
        self.filename = name


class ValuesNotImprovingEarlyStopper:
    """Requests a stop once more than ``patience`` consecutive calls brought no improvement of any of the
    watched ``st_it`` values; ``keys`` holds ``(key, low_is_good)`` pairs."""

    def __init__(self, patience: int, keys: Sequence[Tuple[str, bool]]):
        self.patience, self.keys = patience, tuple(keys)
        self.best_values = [float("inf") if low else float("-inf") for _, low in self.keys]
        self.num_bad_calls = 0

    def __call__(self, st_x: StX, st_it: StIt) -> None:
        improved = False
        for slot, (key, low_is_good) in enumerate(self.keys):
            value, best = st_it[key], self.best_values[slot]
            if (value < best) if low_is_good else (value > best):
                self.best_values[slot] = value
                improved = True
        self.num_bad_calls = 0 if improved else self.num_bad_calls + 1
        if self.num_bad_calls > self.patience:
            st_it["stop"] = True
            getLogger(__name__).info(f"Early stopping at st_it['num_iters_done']={st_it['num_iters_done']}")


class _StopAfter:
    """Hook that raises the stop flag from iteration ``n`` on."""

    def __init__(self, n: int):
        self.n = n

    def __call__(self, st_x: StX, st_it: StIt) -> None:
        st_it["stop"] = st_it["stop"] or st_it["num_iters_done"] >= self.n


def make_stopper_after_n_iters(n: int) -> Callable[[StX, StIt], None]:
    return _StopAfter(n)


class _StopOnNonFiniteLoss:
    """Hook for the ``after_back`` list: on a NaN / infinite loss it raises the stop flag and leaves what is needed
    to reproduce the failure - the model's state_dict (named after the iteration, loss and regulariser term) and the
    batch (``x``, ``y``, ``indices``, ``output``) - in ``dir/nan_loss_stop`` (``nan_loss_stop_rank<r>`` when several
    ranks run: every rank that saw a non-finite loss writes its own).  Same file names as dctn/training.py:213-237."""

    DUMPED = ("x", "y", "indices", "output")

    def __init__(self, dir: str, set_breakpoint: bool):
        self.dir, self.set_breakpoint = dir, set_breakpoint

    def _target(self) -> str:
        leaf = "nan_loss_stop" if _world() == 1 else f"nan_loss_stop_rank{dist.get_rank()}"
        return os.path.join(self.dir, leaf)

    def __call__(self, st_x: StX, st_it: StIt) -> None:
        if bool(torch.isfinite(st_it["loss"])):
            return
        log = getLogger(__name__)
        log.warning("Stopping because of NaN or Inf loss")
        st_it["stop"] = True
        target = self._target()
        if os.path.exists(target):
            log.error(f"subdir={target!r} already exists")
        else:
            os.mkdir(target)
            tag = f"nitd={st_it['num_iters_done']}_loss={st_it['loss']:.3f}_reg_term={st_it['reg_term']:.3f}"
            torch.save(st_x["model"].state_dict(), os.path.join(target, f"model_{tag}.pth"))
            for key in self.DUMPED:
                torch.save(st_it[key], os.path.join(target, key + ".pth"))
        if self.set_breakpoint:
            breakpoint()


def make_stopper_on_nan_loss(dir: str, set_breakpoint: bool) -> Callable[[StX, StIt], None]:
    return _StopOnNonFiniteLoss(dir, set_breakpoint)


def log_parameters_stats(st_x: StX, st_it: StIt) -> None:
    """Mean / std / shape of every parameter, one log line each."""
    log = getLogger(f"{__name__}.log_parameters_stats")
    log.info(f"After {st_it['num_iters_done']:07} iters:")
    with torch.no_grad():
        for name, param in st_x["model"].named_parameters():
            log.info(f"{name}: mu={param.float().mean():.7e}, sigma={param.float().std(unbiased=False):.7e}, "
                     f"shape={tuple(param.shape)}")

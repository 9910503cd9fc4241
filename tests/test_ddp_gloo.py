"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: batch sharding, parameter
broadcast, the flat-bucket gradient all-reduce, and metric reduction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dctn_amd import ddp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = ddp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(1234 + rank)                       # deliberately different init per rank
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    ddp.broadcast_parameters(model.parameters())
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    torch.manual_seed(7)
    x = torch.randn(1, 8, 2, 3)                          # (C, B, ...) layout: batch is dim 1
    y = torch.randn(8, 3)
    xs = ddp.shard_batch(x, rank, world)
    assert xs.shape[1] == 4 and torch.equal(xs, x[:, rank * 4 : rank * 4 + 4])
    loss = ((model(xs[0].reshape(4, 6)) - y[rank * 4 : rank * 4 + 4]) ** 2).sum()
    loss.backward()
    reducer = ddp.FlatGradAllReducer(model.parameters(), average=False)
    reducer()
    # single-process reference on the full batch
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    ref.load_state_dict(model.state_dict())
    ((ref(x[0].reshape(8, 6)) - y) ** 2).sum().backward()
    for p, pr in zip(model.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, pr.grad, atol=1e-5)
    # gradients that already sit back to back in one storage (what the fused EPS + head backward
    # produces) are all-reduced in place through one aliasing view: no gather / scatter
    params = [torch.nn.Parameter(torch.zeros(2, 3)), torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(1))]
    bucket = torch.arange(11, dtype=torch.float32) * (rank + 1)
    params[0].grad, params[1].grad, params[2].grad = bucket[:6].view(2, 3), bucket[6:10], bucket[10:]
    red2 = ddp.FlatGradAllReducer(params, average=True)
    assert red2._contiguous_flat([p.grad for p in params]) is not None
    red2()
    assert torch.allclose(bucket, torch.arange(11, dtype=torch.float32) * 1.5)      # mean of x1 and x2, in place
    assert params[1].grad.data_ptr() == bucket[6:].data_ptr()
    params[1].grad = torch.ones(4)                                                   # no longer one storage: bucket path
    assert red2._contiguous_flat([p.grad for p in params]) is None
    red2()
    assert torch.allclose(params[1].grad, torch.ones(4))
    # evaluation.score: every rank scores its shard, all ranks report the global mean CE / accuracy;
    # training.train_step: identical parameters after an update from different shards
    from dctn_amd import evaluation, training

    torch.manual_seed(99)
    clf = torch.nn.Linear(6, 3)
    ddp.broadcast_parameters(clf.parameters())
    feats, labels = torch.randn(1, 8, 1, 2, 3), torch.randint(0, 3, (8,))
    wrap = lambda xb: clf(xb[0].reshape(xb.shape[1], 6))          # (C, B, ...) input layout
    shard = [(feats[:, rank * 4 : rank * 4 + 2], labels[rank * 4 : rank * 4 + 2], None),
             (feats[:, rank * 4 + 2 : rank * 4 + 4], labels[rank * 4 + 2 : rank * 4 + 4], None)]
    loss_g, acc_g = evaluation.score(wrap, shard, torch.device("cpu"))
    full = clf(feats[0].reshape(8, 6))
    assert abs(loss_g - float(torch.nn.functional.cross_entropy(full, labels))) < 1e-6
    assert abs(acc_g - float((full.argmax(1) == labels).float().mean())) < 1e-9

    class Wrapped(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.clf = clf

        def forward(self, xb):
            return self.clf(xb[0].reshape(xb.shape[1], 6))

    net = Wrapped()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    red3 = ddp.FlatGradAllReducer(net.parameters(), average=True)
    res = training.train_step(net, feats[:, rank * 4 : rank * 4 + 4], labels[rank * 4 : rank * 4 + 4],
                              torch.nn.functional.cross_entropy, opt,
                              reg_fn=lambda m: m.clf.weight.norm() ** 2, reg_coeff=1e-3, reducer=red3)
    assert res["output"].shape == (4, 3)
    flat2 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered2 = [torch.empty_like(flat2) for _ in range(world)]
    dist.all_gather(gathered2, flat2)
    assert all(torch.equal(t, gathered2[0]) for t in gathered2)
    tot, cnt = ddp.all_reduce_scalar_sums(torch.tensor(float(rank + 1)), torch.tensor(4.0))
    assert float(tot) == 3.0 and float(cnt) == 8.0
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_flat_bucket_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def test_single_process_is_a_noop():
    m = torch.nn.Linear(3, 2)
    m(torch.randn(4, 3)).sum().backward()
    g = m.weight.grad.clone()
    ddp.FlatGradAllReducer(m.parameters())()
    assert torch.equal(m.weight.grad, g)
    assert ddp.shard_batch(torch.zeros(1, 9, 2), 1, 2).shape == (1, 4, 2)


# ------------------------------------------------------------------ the reference's training loop, two ranks
def _loop_data():
    g = torch.Generator().manual_seed(21)
    return torch.randn(24, 6, generator=g), torch.randint(0, 3, (24,), generator=g)


def _loop_model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))


def _train_worker(rank, world, port, q, ckpt_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ddp.init_from_env("gloo")
    from dctn_amd import training as T

    x, y = _loop_data()
    # rank r sees samples r, r + world, ... of every batch of 8: batches of 4 per rank
    batches = [(x[b * 8 + rank : b * 8 + 8 : world], y[b * 8 + rank : b * 8 + 8 : world], torch.arange(4)) for b in range(3)]
    model = _loop_model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    seen = []

    def metrics(st_x, st_it):       # what the reference's evaluation hook would put there
        st_it.update(train_acc=0.5, val_acc=0.25 + 0.01 * st_it["num_iters_done"], train_mean_ce=1.0, val_mean_ce=2.0)
        seen.append(st_it["num_iters_done"])

    def rank1_wants_out(st_x, st_it):
        if rank == 1 and st_it["num_iters_done"] == 4:
            st_it["stop"] = True

    keep = T.LastModelsCheckpointer(ckpt_dir, 2)
    st_x, st_it = T.train(batches, model, opt, torch.device("cpu"), torch.nn.functional.cross_entropy,
                          lambda sx, si: sum((p ** 2).sum() for p in sx["model"].parameters()), 1e-3,
                          at_iter_start=[], after_back=[], after_param_upd=[metrics, keep, rank1_wants_out])
    assert st_it["num_iters_done"] == 4 and st_it["stop"] and seen == [0, 1, 2, 3, 4]   # both ranks left together
    q.put((rank, [p.detach().numpy().copy() for p in model.parameters()]))   # by value: the worker may exit first
    dist.barrier()
    dist.destroy_process_group()


def test_reference_training_loop_two_ranks(tmp_path):
    from dctn_amd import training as T

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: [torch.from_numpy(a) for a in arrs] for r, arrs in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for a, b in zip(got[0], got[1]):
        assert torch.equal(a, b)
    # rank 0 alone wrote the checkpoints: the newest two, under the reference's names
    names = sorted(os.listdir(tmp_path))
    assert names == ["model_nitd=0000003_tracc=0.5000_vacc=0.2800_trmce=1.0000_vmce=2.0000.pth",
                     "model_nitd=0000004_tracc=0.5000_vacc=0.2900_trmce=1.0000_vmce=2.0000.pth"]
    state = torch.load(os.path.join(tmp_path, names[1]))
    assert all(torch.equal(state[k], v) for k, v in zip(state, got[0]))
    # one process on the whole batches takes the same 5 steps (mean of the shard gradients == gradient of the mean loss)
    x, y = _loop_data()
    model = _loop_model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    T.train([(x[b * 8 : b * 8 + 8], y[b * 8 : b * 8 + 8], torch.arange(8)) for b in range(3)], model, opt,
            torch.device("cpu"), torch.nn.functional.cross_entropy,
            lambda sx, si: sum((p ** 2).sum() for p in sx["model"].parameters()), 1e-3,
            at_iter_start=[], after_back=[], after_param_upd=[T.make_stopper_after_n_iters(4)])
    for a, p in zip(got[0], model.parameters()):
        assert torch.allclose(a, p.detach(), atol=1e-6)


# ------------------------------------------------------------------ ConvSBS classifier: gradient layout over two ranks
def _sbs_layout_worker(rank, world, port, q):
    """CPU: the model's parameters and the gradient LAYOUT its backward produces (one flat buffer per string, the cores'
    gradients views of it in string order - dctn_amd/conv_sbs.py `_ConvSBSFunction.backward`); the arithmetic itself runs
    on the GPU only (tests/test_gpu_ddp.py)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tests.sbs_classifier import ConvSBSClassifier

    ddp.init_from_env("gloo")
    torch.manual_seed(50 + rank)
    model = ConvSBSClassifier(bond=4)
    ddp.broadcast_parameters(model.parameters())
    flat0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat0) for _ in range(world)]
    dist.all_gather(gathered, flat0)
    assert torch.equal(gathered[0], gathered[1])
    strings = [s for layer in model.layers for s in layer.strings]
    assert len(strings) == 5
    gen = torch.Generator().manual_seed(7 + rank)
    flats = []
    for s in strings:
        n = sum(c.numel() for c in s.cores)
        flat = torch.randn(n, generator=gen)
        flats.append(flat.clone())
        off = 0
        for c in s.cores:
            c.grad = flat[off : off + c.numel()].view_as(c)
            off += c.numel()
    # one string: its gradients ARE one bucket - all-reduced in place, one collective
    red1 = ddp.FlatGradAllReducer(strings[0].parameters(), average=False)
    assert red1._contiguous_flat([c.grad for c in strings[0].cores]) is not None
    # whole model: five buffers - gathered into one bucket, ONE collective, scattered back
    red = ddp.FlatGradAllReducer(model.parameters(), average=True)
    assert red._contiguous_flat([p.grad for p in model.parameters()]) is None
    red()
    # by value (numpy): a tensor travels as a shared-memory handle the parent fetches from THIS process, which may have
    # exited by then (ConnectionResetError in the parent's q.get)
    q.put((rank, [f.numpy().copy() for f in flats],
           [torch.cat([c.grad.reshape(-1) for c in s.cores]).numpy().copy() for s in strings]))
    dist.barrier()
    dist.destroy_process_group()


def test_convsbs_classifier_gradient_buckets_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sbs_layout_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: ([torch.from_numpy(a) for a in f], [torch.from_numpy(a) for a in g]) for r, f, g in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for s in range(5):
        mean = (got[0][0][s] + got[1][0][s]) / 2
        assert torch.allclose(got[0][1][s], mean) and torch.equal(got[0][1][s], got[1][1][s])

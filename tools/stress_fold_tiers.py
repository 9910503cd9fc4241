"""Worst cases of the D = 16 fold at 100 000 windows: every window through the fall-back tiers (no hang, bounded time).
    python tools/stress_fold_tiers.py"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import dctn_amd
from dctn_amd.logmatmulexp import logmatmulexp_fold
dev = torch.device('cuda:0')
torch.manual_seed(0)
for scale, tag in ((1.0, 'benign'), (40.0, 'every window rejected by the factored kernels'), (None, 'one column 60 nats above the others in every matrix (backward: second tier)')):
    m = torch.randn(100000, 9, 16, 16, device=dev)
    if scale is None:
        m[:, :, :, 3] += 60.0
    else:
        m *= scale
    m.requires_grad_(True)
    dy = torch.randn(100000, 16, 16, device=dev)
    for rep in range(2):
        m.grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = logmatmulexp_fold(m)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        y.backward(dy)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{tag}: forward {1e3*(t1-t0):.2f} ms, backward {1e3*(t2-t1):.2f} ms, finite grads {bool(torch.isfinite(m.grad).all())}, finite out {bool(torch.isfinite(y).all())}", flush=True)

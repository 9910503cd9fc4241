#!/usr/bin/env python3
"""Randomised stress of logmatmulexp (direct kernels and the factored exp -> MFMA GEMM -> log path) against float64:
random shapes, dynamic ranges up to +-300, -inf entries, forward + both gradients.  python tools/stress_lme.py [n] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dctn_amd  # noqa: E402
from dctn_amd.logmatmulexp import logmatmulexp_batched  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    kernels = {}
    for case in range(n):
        nb, T, Rr, I = rng.choice([1, 1, 2, 5]), rng.randrange(1, 200), rng.randrange(1, 200), rng.randrange(1, 200)
        torch.manual_seed(case)
        scale = rng.choice([1.0, 3.0, 30.0, 100.0])
        a, b = torch.randn(nb, T, Rr) * scale, torch.randn(nb, Rr, I) * scale
        if rng.random() < 0.4:   # isolated -inf entries (never a whole row / column: the reference's gradient is NaN there)
            for _ in range(5):
                if Rr > 1:
                    a[rng.randrange(nb), rng.randrange(T), rng.randrange(Rr - 1)] = -float("inf")
                    b[rng.randrange(nb), rng.randrange(Rr - 1), rng.randrange(I)] = -float("inf")
        if rng.random() < 0.3:
            a[:, :, 0] += rng.choice([200.0, 300.0])
        ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        y = logmatmulexp_batched(ad, bd)
        kf = dctn_amd.last_kernel()
        a64, b64 = a.double(), b.double()
        want = torch.logsumexp(a64.unsqueeze(3) + b64.unsqueeze(1), dim=2)
        yc = y.detach().cpu().double()
        fin = torch.isfinite(want)
        assert torch.equal(torch.isfinite(yc), fin), (case, "finite pattern", kf)
        err = float(((yc[fin] - want[fin]).abs() / (1.0 + want[fin].abs())).max()) if fin.any() else 0.0
        assert err < 5e-6, (case, "forward", err, kf, (nb, T, Rr, I), scale)
        if fin.all():
            dy = torch.randn(nb, T, I)
            y.backward(dy.to(DEV))
            wgt = torch.exp(a64.unsqueeze(3) + b64.unsqueeze(1) - want.unsqueeze(2)) * dy.double().unsqueeze(2)
            for name, got, ref in (("dA", ad.grad, wgt.sum(3)), ("dB", bd.grad, wgt.sum(1))):
                e = float((got.cpu().double() - ref).abs().max()) / float(ref.abs().max().clamp_min(1.0))
                assert e < 5e-5, (case, name, e, kf, dctn_amd.last_kernel(), (nb, T, Rr, I), scale)
        kernels[kf] = kernels.get(kf, 0) + 1
    print("all ok;", sorted(kernels.items()), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Randomised stress of the two-halves EPS paths (float64 / float32 / bf16) against the CPU oracle: many random shapes
that the dispatcher sends to eps_halves.hip, forward + both gradients, fixed seed.  Not part of the test suite (it takes
a few minutes); run on an MI355X:  python tools/stress_halves.py [ncases] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dctn_amd  # noqa: E402
from dctn_amd.eps import eps  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

DEV = torch.device("cuda:0")
TOL = {torch.float64: 1e-9, torch.float32: 4e-4, torch.bfloat16: 3e-2}


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    done, kernels = 0, {}
    while done < ncases:
        C, K, Q = rng.choice([1, 1, 2, 3]), rng.choice([1, 2, 2, 3, 4]), rng.choice([2, 2, 3, 4, 5, 8])
        N = K * K * C
        O = rng.choice([1, 2, 3, 4, 5, 6, 8, 10, 16])
        if N < 2 or N > 18 or Q**N * O < 2048 or Q**N * O > 2**19:
            continue
        H, W = K + rng.randrange(0, 8), K + rng.randrange(0, 8)
        B = rng.choice([1, 2, 3, 7])
        if B * (H - K + 1) * (W - K + 1) < 64 or B * (H - K + 1) * (W - K + 1) * Q**N * O > 1.5e9:
            continue
        dtype = rng.choice([torch.float64, torch.float32, torch.bfloat16])
        torch.manual_seed(done)
        x = torch.randn(C, B, H, W, Q).to(dtype)
        core = (torch.randn(*(Q,) * N, O) * Q ** (-N / 4)).to(dtype)
        xd = x.to(DEV)
        if rng.random() < 0.3:
            xd = xd.permute(0, 1, 3, 2, 4).contiguous().permute(0, 1, 3, 2, 4)
        xd, cd = xd.requires_grad_(True), core.to(DEV).requires_grad_(True)
        y = eps(cd, xd)
        kf = dctn_amd.last_kernel()
        want = R.eps_4step(core.double(), x.double())
        dy = torch.randn(*want.shape).to(dtype)
        y.backward(dy.to(DEV))
        kb = dctn_amd.last_kernel()
        gc, gx = R.grads(R.eps_4step, [core.double(), x.double()], dy.double())
        for name, got, ref in (("forward", y, want), ("dX", xd.grad, gx), ("dCore", cd.grad, gc)):
            err = float((got.detach().cpu().double() - ref).abs().max()) / (float(ref.abs().max()) or 1.0)
            if not err <= TOL[dtype]:
                print(f"FAIL case {done}: C={C} K={K} Q={Q} O={O} B={B} {H}x{W} {dtype} {name} err={err:.3e} [{kf} / {kb}]", flush=True)
                sys.exit(1)
        kernels[(str(dtype).split(".")[1], kf)] = kernels.get((str(dtype).split(".")[1], kf), 0) + 1
        done += 1
        if done % 20 == 0:
            print(f"{done} cases ok", flush=True)
    print("all ok;", sorted(kernels.items()), flush=True)


if __name__ == "__main__":
    main()

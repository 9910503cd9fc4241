"""A/B of a per-call option flag (DCTN_OPT_SMALL_CHUNKS) on the register families (python tools/q2f32_ab.py 1024 f32|bf16): raw C-ABI calls of the fused forward and backward, timed as
chains of dependent launches from a HIP graph (bench.eps_call_timers)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dctn_amd import _lib as L
from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
DT = {"f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
torch.manual_seed(0)
model = EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, dev, DT, image_size=28)
x = bench.synthetic_input(B, 28, 2, DT, dev, 1)
core = model.epses[0].detach().contiguous()
w, b = model.linear.weight.detach().contiguous(), model.linear.bias.detach().contiguous()

def chain(fn, n=20):
    def body():
        for _ in range(n):
            fn()
    return bench.device_time(body, dev, 10) / n * 1e6

REP = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for name, opt in (("default", 0), ("flag", L.OPT_SMALL_CHUNKS)) * REP:
    L._options = opt
    t = bench.eps_call_timers(core, x, False, dev, head=(w, b))
    pol = L.precision()
    print(f"{name:8s} head_fwd {chain(t['head_fwd']):6.2f} us   bwd call {chain(t['bwd']):6.2f} us   dcore kernel "
          f"{chain(lambda: t['bwd'](pol | L.OPT_MAIN_KERNEL_ONLY)):6.2f} us")
L._options = 0

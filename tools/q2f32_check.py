"""Quick device check + timing of the exact-f32 register family (cfg2 in float32): parity vs the oracle, then the
fwd + bwd step replayed from a HIP graph."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dctn_amd
from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd
from oracle import ref_cpu as R

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, dev, torch.float32, image_size=28)
u = torch.rand(1, B, 28, 28)
x = torch.stack((torch.sin(u * torch.pi / 2) ** 2, torch.cos(u * torch.pi / 2) ** 2), dim=-1).to(dev)
g = torch.randn(B, 10, device=dev)
out = model(x)
print("fwd kernel", dctn_amd.last_kernel())
out.backward(g)
print("bwd kernel", dctn_amd.last_kernel())
nb = min(B, 16)
c64 = model.epses[0].detach().cpu().double().requires_grad_(True)
w64 = model.linear.weight.detach().cpu().double().requires_grad_(True)
b64 = model.linear.bias.detach().cpu().double().requires_grad_(True)
want = R.eps_plus_linear_forward([c64], w64, b64, x[:, :nb].cpu().double())
print("out rel err", float((out[:nb].detach().cpu().double() - want).abs().max() / want.abs().max()))
if nb == B:
    want.backward(g.cpu().double())
    for n, a, b in (("dCore", model.epses[0].grad, c64.grad), ("dW", model.linear.weight.grad, w64.grad), ("dB", model.linear.bias.grad, b64.grad)):
        print(n, "rel err", float((a.cpu().double() - b).abs().max() / b.abs().max()))

del out, want
import bench

def step():
    for p in model.parameters():
        p.grad = None
    model(x).backward(g)

def fwd():
    with torch.no_grad():
        model(x)

def ten(fn):
    def body():
        for _ in range(10):
            fn()
    return body

t_fb = bench.device_time(ten(step), dev, 20) / 10
t_f = bench.device_time(ten(fwd), dev, 20) / 10
print(f"B={B}: {t_fb * 1e6:.1f} us per fwd+bwd step, {t_f * 1e6:.1f} us per forward (graph, 10 per launch)")

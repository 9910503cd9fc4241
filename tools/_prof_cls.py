import sys, torch
sys.path.insert(0, '/root/repo')
from tests.sbs_classifier import ConvSBSClassifier
dev = torch.device('cuda:0')
bond = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
m = ConvSBSClassifier(bond=bond, reference_form=True).to(dev)   # the reference's own model (mnist.py:255-283)
x = torch.rand(1, 128, 28, 28, 2, device=dev)
m.calibrate(x)
y = torch.randint(0, 10, (128,), device=dev)
for _ in range(30):
    for p in m.parameters(): p.grad = None
    torch.nn.functional.cross_entropy(m(x), y).backward()
torch.cuda.synchronize()

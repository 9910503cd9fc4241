import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time, dctn_amd
from dctn_amd.eps import eps
dev=torch.device("cuda")
K,Q,O=3,4,6
core=torch.randn(*(Q,)*(K*K), O, device=dev)*Q**(-4.5)
for B in (32, 64, 128, 256, 512):
    x=torch.rand(1,B,25,25,Q,device=dev)+0.1
    with torch.no_grad():
        for _ in range(3): y=eps(core,x)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): y=eps(core,x)
        e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    Wn=B*23*23
    fl=2*Q**9*O*Wn
    print(B, Wn, dctn_amd.last_kernel(), "%.3f ms  %.1f TFLOP/s  frac %.2f" % (ms, fl/ms/1e9, fl/ms/1e9/157.3))

import sys, torch
sys.path.insert(0, '/root/repo')
from tests.sbs_classifier import ConvSBSClassifier
from dctn_amd.conv_sbs import matrix_core_sweep
import contextlib
dev = torch.device('cuda:0')
# both forms of tests/sbs_classifier.py: the reference's own model (mnist.py:255-283: layers + mean, strings rescaled on a
# batch) and the tests' variant with tanh(scale * output) between the layers (20 more elementwise launches per step)
ONLY = [int(a) for a in sys.argv[1:]]   # e.g. `python tools/time_sbs_classifier.py 16` (under rocprofv3: one bond's kernels)
for bond, ref in ((2, True), (4, True), (8, True), (16, True), (2, False), (4, False)):
    if ONLY and (bond not in ONLY or not ref):
        continue
    torch.manual_seed(0)
    m = ConvSBSClassifier(bond=bond, reference_form=ref).to(dev)
    x = torch.rand(1, 128, 28, 28, 2, device=dev)
    m.calibrate(x)
    y = torch.randint(0, 10, (128,), device=dev)
    def step():
        for p in m.parameters(): p.grad = None
        torch.nn.functional.cross_entropy(m(x), y).backward()
    for name, ctx in (("string by string, matrix-core sweep (round 2 path)", matrix_core_sweep), ("default (strings of a layer in one launch)", contextlib.nullcontext)):
        if ref and ctx is matrix_core_sweep:
            continue
        with ctx():
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                step(); step()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): g.replay()
            e1.record(); torch.cuda.synchronize()
            print(f"bond {bond} ({'reference form' if ref else 'tanh between layers'}): {name}: {e0.elapsed_time(e1)/50*1e3:.1f} us per training forward+backward (B=128, 28x28, three layers)", flush=True)

"""Host-visible fixed cost of one graph replay bracketed by synchronize (what each timed block of bench.py carries)."""
import time, torch
dev = torch.device("cuda:0")
a = torch.zeros(1024, device=dev)
def bracket(fn, n=200):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6
for nk in (1, 60):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        a.add_(1)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(nk):
            a.add_(1)
    print(f"graph of {nk} tiny kernels: replay + synchronize = {bracket(g.replay):.1f} us (median)")
print(f"one eager tiny kernel + synchronize = {bracket(lambda: a.add_(1)):.1f} us")
print(f"synchronize alone = {bracket(lambda: None):.1f} us")

"""Does a hipMemsetAsync issued during stream capture become a node that runs on every replay?"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")
for n in (1000, 1152 // 4 * 4, 300000):
    buf = torch.ones(n, device=dev)
    other = torch.zeros(16, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        other.add_(1)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, n * 4, torch.cuda.current_stream().cuda_stream)
        other.add_(1)
    res = []
    for rep in range(3):
        buf.fill_(5.0)
        g.replay(); torch.cuda.synchronize()
        res.append(float(buf.abs().max()))
    print("n", n, "rc", rc, "max after replays", res)

"""HBM write / copy bandwidth on this box (torch kernels, captured graph of 20 ops)."""
import torch
def timeit(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); g.replay(); b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for mb in (10, 41, 164, 656):
    n = mb * 1024 * 1024 // 4
    x, y = torch.zeros(n, device="cuda"), torch.ones(n, device="cuda")
    tf = timeit(lambda: x.fill_(2.0)); tc = timeit(lambda: x.copy_(y)); tr = timeit(lambda: y.sum())
    print(f"{mb:4d} MB  fill {tf:7.2f} us = {mb * 1.048576 / tf:5.2f} TB/s | copy {tc:7.2f} us = {2 * mb * 1.048576 / tc:5.2f} TB/s (r+w) | sum {tr:7.2f} us = {mb * 1.048576 / tr:5.2f} TB/s")

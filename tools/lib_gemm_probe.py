"""What does the vendor library (torch.mm -> hipBLASLt / rocBLAS, fp32) reach on the hot-path GEMM shapes?
Reference point only; the product path uses csrc/gemm.hip."""
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
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [("head fwd  NN", 960, 5004, 512, False, False), ("head dW   TN", 512, 5004, 960, True, False),
          ("head dX   NT", 960, 512, 5004, False, True), ("xproj     NN", 1024, 2048, 512, False, False),
          ("lstm dW   TN", 512, 2048, 1024, True, False), ("lstm dX   NT", 1024, 512, 2048, False, True),
          ("enc fwd   NN", 64, 512, 20000, False, False), ("enc dW    TN", 20000, 512, 64, True, False),
          ("big       NN", 4096, 4096, 2048, False, False)]
for name, M, N, K, tA, tB in shapes:
    A = torch.randn((K, M) if tA else (M, K), device="cuda")
    Bm = torch.randn((N, K) if tB else (K, N), device="cuda")
    C = torch.zeros(M, N, device="cuda")
    a = A.t() if tA else A
    b = Bm.t() if tB else Bm
    t = timeit(lambda: torch.mm(a, b, out=C))
    print(f"{name} M={M:5d} N={N:5d} K={K:5d}: {t:8.2f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF")

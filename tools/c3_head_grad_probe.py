"""Config-3 vocabulary-head gradient GEMMs (H = 256): vendor sgemm against csrc/gemm.hip with split-K, each as a
captured graph of 20 launches.  dX = dlogits[960x5001] @ Wo^T (NT, long K, skinny output) is where rocBLAS' pick is
poor (43 TF)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import masters_thesis_amd.ops as ops
be = ops.backend()
f = lambda *s: torch.randn(*s, device="cuda")


def timeit(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); g.replay(); b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


Bt, V, ldV = 960, 5001, 5004
for H in (256, 512):
    dlog, Wo, X = f(Bt, ldV), f(H, ldV), f(Bt, H)
    dX, dW = torch.zeros(Bt, H, device="cuda"), torch.zeros(H, ldV, device="cuda")
    work = torch.empty(64 * max(Bt * H, H * ldV), device="cuda")
    fl = 2.0 * Bt * V * H
    t = timeit(lambda: be.gemm_blas(dlog, Wo, dX, Bt, H, V, ldV, ldV, H, transB=True))
    print(f"H={H} dX NT {Bt}x{H}x{V}: rocBLAS {t:6.1f} us {fl / t / 1e6:6.1f} TF")
    for sk in (4, 8, 12, 16, 24, 32):
        t = timeit(lambda: be.gemm(dlog, Wo, dX, Bt, H, V, ldV, ldV, H, transB=True, splitk=sk, work=work))
        print(f"      gemm.hip splitk={sk:2d}: {t:6.1f} us {fl / t / 1e6:6.1f} TF")
    t = timeit(lambda: be.gemm_blas(X, dlog, dW, H, V, Bt, H, ldV, ldV, transA=True))
    print(f"H={H} dW TN {H}x{V}x{Bt}: rocBLAS {t:6.1f} us {fl / t / 1e6:6.1f} TF")
    for sk in (1, 2, 4):
        t = timeit(lambda: be.gemm(X, dlog, dW, H, V, Bt, H, ldV, ldV, transA=True, splitk=sk, work=work))
        print(f"      gemm.hip splitk={sk:2d}: {t:6.1f} us {fl / t / 1e6:6.1f} TF")

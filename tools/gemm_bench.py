"""Times the hot-path GEMM shapes for every workgroup tile (calibrates the tile heuristic)."""
import ctypes as C, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from masters_thesis_amd import _lib
lib = _lib.load()
P, I32, F32 = C.c_void_p, C.c_int32, C.c_float
lib.tnt_gemm_f32_tile.argtypes = [P]*5 + [I32]*9 + [F32, I32, I32, P, I32, I32, P]
lib.tnt_gemm_f32_tile.restype = I32
shapes = [  # name, M, N, K, tA, tB
    ("head fwd NN", 960, 5001, 512, 0, 0), ("head dW TN", 512, 5001, 960, 1, 0), ("head dX NT", 960, 512, 5001, 0, 1),
    ("xproj NN", 1024, 2048, 512, 0, 0), ("dU TN", 512, 2048, 1024, 1, 0), ("dXin NT", 1024, 512, 2048, 0, 1),
    ("enc fwd NN", 64, 512, 20000, 0, 0), ("enc dW TN", 20000, 512, 64, 1, 0), ("out fwd NN(c3)", 960, 5001, 256, 0, 0),
]
s = torch.cuda.current_stream().cuda_stream
for name, M, N, K, tA, tB in shapes:
    A = torch.randn((K, M) if tA else (M, K), device="cuda")
    ldb = (K if tB else N); ldb4 = (ldb + 3) // 4 * 4
    Bm = torch.zeros((N if tB else K), ldb4, device="cuda"); Bm[:, :ldb].normal_()
    ldc = (N + 3) // 4 * 4
    Cm = torch.zeros(M, ldc, device="cuda")
    lda = A.shape[1]
    res = []
    for sk in (1, 2, 4, 8, 16, 32, 64):
        if sk > 1 and K // sk < 64: continue
        work = torch.zeros(sk * M * N, device="cuda") if sk > 1 else None
        for bm, bn in ((128, 128), (64, 128), (128, 64), (64, 64)):
            def run():
                rc = lib.tnt_gemm_f32_tile(A.data_ptr(), Bm.data_ptr(), Cm.data_ptr(), None, None, M, N, K, lda, ldb4, ldc,
                                           tA, tB, 0, 0.2, 0, sk, work.data_ptr() if work is not None else None, bm, bn, s)
                assert rc == 0
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            res.append((us, sk, bm, bn))
    res.sort()
    best = res[0]
    print(f"{name:16s} M={M} N={N} K={K}: best {best[0]:7.1f} us ({2*M*N*K/best[0]/1e6:6.1f} TF) sk={best[1]} tile={best[2]}x{best[3]} | " +
          " ".join(f"[{u:.0f}us sk{k} {a}x{b}]" for u, k, a, b in res[1:6]))

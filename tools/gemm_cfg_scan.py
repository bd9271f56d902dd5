"""Times every one-round configuration of tnt_gemm_fused_f32 on the hot-path shapes next to the vendor sgemm
(tnt_gemm_blas_f32) and checks each result against torch.matmul.  Calibrates pick_cfg_1r in csrc/gemm.hip."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import masters_thesis_amd.ops as ops
be = ops.backend()
NCFG = 12
shapes = [  # name, M, N, K, tA, tB, batch, colsum
    ("head fwd NN", 960, 5001, 512, 0, 0, 1, 0), ("c3 head fwd NN", 960, 5001, 256, 0, 0, 1, 0),
    ("xproj NN", 1024, 2048, 512, 0, 0, 1, 0), ("head dW TN", 512, 5001, 960, 1, 0, 1, 1), ("head dX NT", 960, 512, 5001, 0, 1, 1, 0),
    ("dU TN", 512, 2048, 1024, 1, 0, 1, 1), ("dU+dW TN x2", 512, 2048, 1024, 1, 0, 2, 1), ("dXin NT", 1024, 512, 2048, 0, 1, 1, 0),
    ("c3 head dW TN", 256, 5001, 960, 1, 0, 1, 1), ("c3 head dX NT", 960, 256, 5001, 0, 1, 1, 0),
    ("c3 inter dW TN", 512, 256, 960, 1, 0, 1, 1), ("c3 inter dX NT", 960, 512, 256, 0, 1, 1, 0),
    ("c3 xproj NN", 960, 2048, 544, 0, 0, 1, 0), ("c3 dW TN", 544, 2048, 960, 1, 0, 1, 1), ("c3 dXin NT", 960, 544, 2048, 0, 1, 1, 0),
]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(a in s[0] for a in sys.argv[1:])]


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


r4 = lambda n: (n + 3) // 4 * 4
for name, M, N, K, tA, tB, batch, cs in shapes:
    lda = r4(M if tA else K); ldb = r4(K if tB else N); ldc = r4(N)
    A = torch.zeros((K if tA else M), lda, device="cuda"); A[:, :(M if tA else K)].normal_()
    A2 = torch.zeros_like(A); A2[:, :(M if tA else K)].normal_()
    Bm = torch.zeros((N if tB else K), ldb, device="cuda"); Bm[:, :(K if tB else N)].normal_()
    Cm, C2 = torch.zeros(M, ldc, device="cuda"), torch.zeros(M, ldc, device="cuda")
    col = torch.zeros(ldc, device="cuda")
    opA = (A[:, :M].t() if tA else A[:, :K]).double(); opA2 = (A2[:, :M].t() if tA else A2[:, :K]).double()
    opB = (Bm[:, :K].t() if tB else Bm[:, :N]).double()
    want, want2 = opA @ opB, opA2 @ opB
    t_blas = timeit(lambda: be.gemm_blas(A, Bm, Cm, M, N, K, lda, ldb, ldc, transA=bool(tA), transB=bool(tB)))
    if batch == 2: t_blas *= 2
    res = []
    for cfg in (1, 9, 21, 22, 24, 25, 26, 28):
        Cm.zero_(); C2.zero_(); col.zero_()
        try:
            run = lambda: be.gemm_fused(A, Bm, Cm, M, N, K, lda, ldb, ldc, transA=bool(tA), transB=bool(tB),
                                        colsum=col if cs else None, A2=A2 if batch == 2 else None, C2=C2 if batch == 2 else None, cfg=cfg)
            us = timeit(run)
        except Exception as e:
            res.append((1e9, cfg, str(e)[-12:])); continue
        err = (Cm[:, :N].double() - want).abs().max().item() / want.abs().max().item()
        if batch == 2: err = max(err, (C2[:, :N].double() - want2).abs().max().item() / want2.abs().max().item())
        if cs: err = max(err, (col[:N].double() - opB.sum(0)).abs().max().item() / opB.sum(0).abs().max().item())
        res.append((us, cfg, "ok" if err < 1e-5 else f"ERR {err:.1e}"))
    res.sort()
    auto = be.gemm_fused_cfg(M, N, K, bool(tA), bool(tB), batch)
    fl = 2.0 * M * N * K * batch
    print(f"{name:16s} {M}x{N}x{K} blas {t_blas:6.1f} us ({fl/t_blas/1e6:5.1f} TF) auto=cfg{auto} | " +
          " ".join(f"[cfg{c} {u:.1f}us {fl/u/1e6:.0f}TF {st}]" for u, c, st in res[:6]))

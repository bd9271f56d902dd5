"""Every epilogue-free GEMM shape one train step routes to the vendor library (ModelBase.gemm_sk), timed both ways:
rocBLAS sgemm against csrc/gemm.hip with the calibrated split-K, each as a captured graph of 20 launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from masters_thesis_amd.model_base import ModelBase

workload = sys.argv[1] if len(sys.argv) > 1 else "dense"
dev = torch.device("cuda", 0)
model = bench.make_model(workload, dev, None)
model.use_graph = False
batch, _ = bench.synth(0, dev)
seen = {}
orig = ModelBase.gemm_sk


def spy(self, A, B, C, M, N, K, lda, ldb, ldc, ws=0, **kw):
    plain = kw.get("bias") is None and kw.get("pre") is None and kw.get("act", 0) == 0
    if plain:
        seen.setdefault((M, N, K, bool(kw.get("transA", False)), bool(kw.get("transB", False))), (lda, ldb, ldc))
    return orig(self, A, B, C, M, N, K, lda, ldb, ldc, ws=ws, **kw)


ModelBase.gemm_sk = spy
model.train_step(batch)
torch.cuda.synchronize()
ModelBase.gemm_sk = orig
be = model.be


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


print(f"# {workload}: {len(seen)} epilogue-free GEMM shapes per step")
for (M, N, K, tA, tB), (lda, ldb, ldc) in sorted(seen.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2]):
    A = torch.randn((K if tA else M), lda, device="cuda")
    Bm = torch.randn((N if tB else K), ldb, device="cuda")
    C = torch.zeros(M, ldc, device="cuda")
    sk = ModelBase.pick_splitk(M, N, K)
    work = torch.empty(max(1, sk * M * N), device="cuda")
    tb = timeit(lambda: be.gemm_blas(A, Bm, C, M, N, K, lda, ldb, ldc, transA=tA, transB=tB))
    th = timeit(lambda: be.gemm(A, Bm, C, M, N, K, lda, ldb, ldc, transA=tA, transB=tB, splitk=sk, work=work))
    fl = 2.0 * M * N * K
    flag = "  <-- tiled kernel faster" if th < 0.92 * tb else ""
    print(f"{'T' if tA else 'N'}{'T' if tB else 'N'} {M:6d} x {N:5d} x {K:6d}: rocBLAS {tb:7.1f} us {fl / tb / 1e6:6.1f} TF | "
          f"gemm.hip splitk={sk:2d} {th:7.1f} us {fl / th / 1e6:6.1f} TF{flag}")

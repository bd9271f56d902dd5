"""Runs one GEMM shape/tile repeatedly (for rocprofv3 --pmc passes)."""
import ctypes as C, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from masters_thesis_amd import _lib
lib = _lib.load()
P, I32, F32 = C.c_void_p, C.c_int32, C.c_float
lib.tnt_gemm_f32_tile.argtypes = [P]*5 + [I32]*9 + [F32, I32, I32, P, I32, I32, P]
M, N, K, tA, tB, bm, bn = [int(v) for v in sys.argv[1:8]]
A = torch.randn((K, M) if tA else (M, K), device="cuda")
ldb = (K if tB else N); ldb4 = (ldb + 3) // 4 * 4
Bm = torch.zeros((N if tB else K), ldb4, device="cuda"); Bm[:, :ldb].normal_()
ldc = (N + 3) // 4 * 4
Cm = torch.zeros(M, ldc, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(10):
    lib.tnt_gemm_f32_tile(A.data_ptr(), Bm.data_ptr(), Cm.data_ptr(), None, None, M, N, K, A.shape[1], ldb4, ldc, tA, tB, 0, 0.2, 0, 1, None, bm, bn, s)
torch.cuda.synchronize()

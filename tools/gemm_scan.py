"""Scan tile-count / K to separate per-workgroup efficiency from occupancy effects."""
import ctypes as C, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from masters_thesis_amd import _lib
lib = _lib.load()
P, I32, F32 = C.c_void_p, C.c_int32, C.c_float
lib.tnt_gemm_f32_tile.argtypes = [P]*5 + [I32]*9 + [F32, I32, I32, P, I32, I32, P]
s = torch.cuda.current_stream().cuda_stream
def t(M, N, K, tA, tB, bm, bn):
    A = torch.randn((K, M) if tA else (M, K), device="cuda")
    Bm = torch.randn((N, K) if tB else (K, N), device="cuda")
    Cm = torch.zeros(M, N, device="cuda")
    run = lambda: lib.tnt_gemm_f32_tile(A.data_ptr(), Bm.data_ptr(), Cm.data_ptr(), None, None, M, N, K, A.shape[1], Bm.shape[1], N, tA, tB, 0, 0.2, 0, 1, None, bm, bn, s)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    return us, 2.0 * M * N * K / us / 1e6
for bm, bn in ((64, 64), (128, 128), (128, 64)):
    for K in (512, 4096):
        for tiles in (256, 512, 1024, 2048, 4096):
            M = 1024 if bm == 64 else 2048
            N = tiles * bm * bn // M
            us, tf = t(M, N, K, 0, 0, bm, bn)
            print(f"tile {bm}x{bn} K={K:5d} tiles={tiles:5d} (M={M},N={N}): {us:8.1f} us {tf:6.1f} TF")

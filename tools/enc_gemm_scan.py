"""Encoder GEMM shapes of config 2: tile / split-K sensitivity, timed in a captured graph of 20 launches."""
import ctypes as C, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from masters_thesis_amd import _lib
lib = _lib.load()
P, I32, F32 = C.c_void_p, C.c_int32, C.c_float
lib.tnt_gemm_f32_tile.argtypes = [P]*5 + [I32]*9 + [F32, I32, I32, P, I32, I32, P]

def timeit(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn(s.cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(n):
                fn(s.cuda_stream)
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); g.replay(); b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

B, N, E = 64, 20000, 512
X = torch.randn(B, N, device="cuda"); W = torch.randn(N, E, device="cuda") * 0.01
Y = torch.zeros(B, E, device="cuda"); dpre = torch.randn(B, E, device="cuda"); dW = torch.zeros(N, E, device="cuda")
work = torch.zeros(256 * B * E + 16, device="cuda")
print("encoder fwd  Y[64x512] = X[64x20000] @ W[20000x512]")
for sk in (16, 32, 64, 128, 256):
    for bm, bn in ((64, 64), (64, 128)):
        t = timeit(lambda s: lib.tnt_gemm_f32_tile(X.data_ptr(), W.data_ptr(), Y.data_ptr(), None, None, B, E, N, N, E, E, 0, 0, 0, 0.2, 0, sk, work.data_ptr(), bm, bn, s))
        print(f"  splitk={sk:4d} tile {bm}x{bn}: {t:7.2f} us")
print("encoder dW   dW[20000x512] = X^T[20000x64] @ dpre[64x512]")
for bm, bn in ((64, 64), (64, 128), (128, 64), (128, 128)):
    t = timeit(lambda s: lib.tnt_gemm_f32_tile(X.data_ptr(), dpre.data_ptr(), dW.data_ptr(), None, None, N, E, B, N, E, E, 1, 0, 0, 0.2, 0, 1, None, bm, bn, s))
    print(f"  tile {bm}x{bn}: {t:7.2f} us   ({N * E * 4 / t / 1e6:.2f} TB/s of output)")
import masters_thesis_amd.ops as ops
be = ops.backend()
lib.tnt_dense_dw_skinny_f32.argtypes = [P, P, P, I32, I32, I32, I32, P]
t = timeit(lambda s: lib.tnt_dense_dw_skinny_f32(X.data_ptr(), dpre.data_ptr(), dW.data_ptr(), N, E, B, N, s))
print(f"  skinny dW kernel: {t:7.2f} us   ({N * E * 4 / t / 1e6:.2f} TB/s of output)")
XT = X.t().contiguous()
for bm, bn in ((64, 64), (64, 128), (128, 128)):
    t = timeit(lambda s: lib.tnt_gemm_f32_tile(XT.data_ptr(), dpre.data_ptr(), dW.data_ptr(), None, None, N, E, B, B, E, E, 0, 0, 0, 0.2, 0, 1, None, bm, bn, s))
    print(f"  NN on pre-transposed X^T, tile {bm}x{bn}: {t:7.2f} us")
Xs = torch.randn(B, 2048, device="cuda")      # small-N control: same kernel, X rows 8 KB apart
dWs = torch.zeros(2048, E, device="cuda")
t = timeit(lambda s: lib.tnt_dense_dw_skinny_f32(Xs.data_ptr(), dpre.data_ptr(), dWs.data_ptr(), 2048, E, B, 2048, s))
print(f"  skinny dW kernel, N=2048: {t:7.2f} us")

import ctypes as C, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from masters_thesis_amd import _lib
lib = _lib.load()
P, I32 = C.c_void_p, C.c_int32
lib.tnt_dense_dw_skinny_f32.argtypes = [P, P, P, I32, I32, I32, I32, P]
def timeit(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn(s.cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(n): fn(s.cuda_stream)
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); g.replay(); b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B, N, E = 64, 20000, 512
X = torch.randn(B, N, device="cuda"); dpre = torch.randn(B, E, device="cuda"); dW = torch.zeros(N, E, device="cuda")
for tpw in (1, 2, 4):
    for grid in (128, 256, 512, 1250):
        os.environ["TNT_DW_GRID"], os.environ["TNT_DW_TPW"] = str(grid), str(tpw)
        t = timeit(lambda s: lib.tnt_dense_dw_skinny_f32(X.data_ptr(), dpre.data_ptr(), dW.data_ptr(), N, E, B, N, s))
        print(f"tpw={tpw} grid.x={grid}: {t:7.2f} us")

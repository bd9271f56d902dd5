import ctypes as C, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from masters_thesis_amd import _lib
lib = _lib.load()
P, I32 = C.c_void_p, C.c_int32
lib.tnt_dense_dw_skinny_f32.argtypes = [P, P, P, I32, I32, I32, I32, P]
B, N, E = 64, 20000, 512
X = torch.randn(B, N, device="cuda"); dpre = torch.randn(B, E, device="cuda"); dW = torch.zeros(N, E, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(10):
    lib.tnt_dense_dw_skinny_f32(X.data_ptr(), dpre.data_ptr(), dW.data_ptr(), N, E, B, N, s)
torch.cuda.synchronize()

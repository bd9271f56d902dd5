"""Ablation timing of the fused LSTM step kernels (B=64, U=512)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import masters_thesis_amd.ops as ops
be = ops.backend()
B, U, T = 64, 512, 16
f = lambda *s: torch.randn(*s, device="cuda") * 0.1
xz, Ur, gates = f(T, B, U, 4), f(U, U, 4), torch.rand(T, B, U, 4, device="cuda")
H, C = f(T + 1, B, U), f(T + 1, B, U)
dZ, dOut = f(T + 1, B, U, 4), f(T, B, U)
da, dc, dout = f(B, U), f(B, U), f(B, U)
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
def fwd_chain():
    for t in range(T):
        be.lstm_step_fwd(xz[t], H[t], C[t], Ur, None, None, 0, None, 0, 0, None, H[t + 1], C[t + 1], None, gates[t], B, U)
def bwd_chain(gemm=True):
    for t in range(T - 1, -1, -1):
        be.lstm_step_bwd(dZ[t + 1] if gemm else None, Ur, da, None, dc, None, dOut[t], None, 0, 0, gates[t], C[t + 1], C[t], dZ[t], da, dc, None, B, U)
g = torch.cuda.CUDAGraph()
for name, fn in (("fwd chain (16 steps)", fwd_chain), ("bwd chain (16 steps)", bwd_chain), ("bwd chain no-gemm", lambda: bwd_chain(False))):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    us = timeit(g.replay)
    print(f"{name:24s}: {us:8.1f} us per chain = {us / T:6.2f} us/step (graph replay)")

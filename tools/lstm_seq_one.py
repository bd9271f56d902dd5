"""Runs the two persistent LSTM chain kernels a few times (for rocprofv3 --pmc passes): fwd and bwd at B=64, U=512, T=15."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import masters_thesis_amd.ops as ops
be = ops.backend()
B, U, T = 64, 512, 15; S = T + 1
rng = np.random.default_rng(1)
f = lambda *s: torch.tensor(rng.standard_normal(s), dtype=torch.float32, device="cuda")
xz, Ur, bl = f(S, B, U, 4) * 0.5, f(U, U, 4) * 0.05, f(U, 4) * 0.1
cap = rng.integers(1, 50, (B, T)).astype(np.int32)
for b in range(B): cap[b, rng.integers(3, T):] = 0
capd = torch.tensor(cap, device="cuda")
Hs, Cs = torch.zeros(S + 1, B, U, device="cuda"), torch.zeros(S + 1, B, U, device="cuda")
Out, G = torch.zeros(T, B, U, device="cuda"), torch.zeros(S, B, U, 4, device="cuda")
dOut, dZ = f(T, B, U) * 0.1, torch.zeros(S, B, U, 4, device="cuda")
sync = torch.zeros(1025, dtype=torch.int32, device="cuda")
work = torch.zeros(be.lstm_seq_bwd_work_floats(B, U), device="cuda")
assert be.lstm_seq_supported(B, U)
for _ in range(10):
    be.lstm_seq_fwd(xz, Hs, Cs, Ur, bl, capd, T, 1, Out, G, S, B, U, sync)
    be.lstm_seq_bwd(Ur, dOut, capd, T, 1, G, Cs, dZ, work, S, B, U, sync)
torch.cuda.synchronize()
assert int(sync[1024]) == 0

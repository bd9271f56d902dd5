import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from masters_thesis_amd import _lib
if os.environ.get("TNT_LIB"): _lib.LIB_PATH = os.environ["TNT_LIB"]
import masters_thesis_amd.ops as ops
be = ops.backend()
B, U, T = 64, 512, 15; S = T + 1
print("supported", be.lstm_seq_supported(B, U))
rng = np.random.default_rng(1)
f = lambda *s: torch.tensor(rng.standard_normal(s), dtype=torch.float32, device="cuda")
xz, Ur, bl = f(S, B, U, 4) * 0.5, f(U, U, 4) * 0.05, f(U, 4) * 0.1
cap = rng.integers(1, 50, (B, T)).astype(np.int32)
for b in range(B): cap[b, rng.integers(3, T):] = 0
capd = torch.tensor(cap, device="cuda")
h0, c0 = f(B, U) * 0.3, f(B, U) * 0.3
def alloc():
    Hs, Cs = torch.zeros(S + 1, B, U, device="cuda"), torch.zeros(S + 1, B, U, device="cuda")
    Hs[0], Cs[0] = h0, c0
    return Hs, Cs, torch.full((T, B, U), 9.0, device="cuda"), torch.zeros(S, B, U, 4, device="cuda")
Hs, Cs, Out, G = alloc()
be.lstm_step_fwd(xz[0], Hs[0], Cs[0], Ur, None, None, 0, None, 0, 0, None, Hs[1], Cs[1], None, G[0], B, U, xz_bias=bl)
for t in range(1, S):
    be.lstm_step_fwd(xz[t], Hs[t], Cs[t], Ur, None, None, 0, capd, T, t - 1, Out[t - 2] if t > 1 else None, Hs[t + 1], Cs[t + 1], Out[t - 1], G[t], B, U, xz_bias=bl)
for rep in range(3):
    Hs2, Cs2, Out2, G2 = alloc()
    sync = torch.zeros(1025, dtype=torch.int32, device="cuda")
    be.lstm_seq_fwd(xz, Hs2, Cs2, Ur, bl, capd, T, 1, Out2, G2, S, B, U, sync)
    torch.cuda.synchronize()
    print("err", int(sync[1024]), "flag0", sync[0:512:64].tolist(), "tickets", sync[512:1024:64].tolist(), "epochs", sync[514:1024:64].tolist())
    for s in range(S + 1):
        d = (Hs[s] - Hs2[s]).abs()
        print(s, "H maxdiff %.3e" % d.max().item(), "rows bad", (d.max(1).values > 0).sum().item(), "cols bad", (d.max(0).values > 0).sum().item(),
              "| G %.3e" % ((G[s - 1] - G2[s - 1]).abs().max().item() if s > 0 else 0.0))
    if rep == 0: print("Out diff %.3e" % (Out - Out2).abs().max().item())


def timeit(fn, n=10):
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


def steps():
    be.lstm_step_fwd(xz[0], Hs[0], Cs[0], Ur, None, None, 0, None, 0, 0, None, Hs[1], Cs[1], None, G[0], B, U, xz_bias=bl)
    for t in range(1, S):
        be.lstm_step_fwd(xz[t], Hs[t], Cs[t], Ur, None, None, 0, capd, T, t - 1, Out[t - 2] if t > 1 else None, Hs[t + 1],
                         Cs[t + 1], Out[t - 1], G[t], B, U, xz_bias=bl)


sync = torch.zeros(1025, dtype=torch.int32, device="cuda")
t_steps = timeit(steps)
t_seq = timeit(lambda: be.lstm_seq_fwd(xz, Hs2, Cs2, Ur, bl, capd, T, 1, Out2, G2, S, B, U, sync))
print(f"16 step launches: {t_steps:7.1f} us ({t_steps / S:.2f} us/step);  persistent: {t_seq:7.1f} us ({t_seq / S:.2f} us/step); err {int(sync[1024])}")


# ---- backward chain: 16 per-step launches vs the persistent push kernel
dOut = f(T, B, U) * 0.1
dZ, dZ2 = torch.zeros(S, B, U, 4, device="cuda"), torch.zeros(S, B, U, 4, device="cuda")
z = lambda: torch.zeros(B, U, device="cuda")
dap, dcp, dop = z(), z(), z()
work = torch.zeros(be.lstm_seq_bwd_work_floats(B, U), device="cuda")


def bsteps():
    for s in range(S - 1, -1, -1):
        first, seq = s == S - 1, s >= 1
        be.lstm_step_bwd(None if first else dZ[s + 1], Ur, None if first else dap, None, None if first else dcp,
                         (None if first else dop) if seq else None, dOut[s - 1] if seq else None, capd if seq else None, T,
                         s - 1 if seq else 0, G2[s], Cs2[s + 1], Cs2[s], dZ[s], dap, dcp, dop if seq else None, B, U)


t_bsteps = timeit(bsteps)
t_bseq = timeit(lambda: be.lstm_seq_bwd(Ur, dOut, capd, T, 1, G2, Cs2, dZ2, work, S, B, U, sync))
torch.cuda.synchronize()
print(f"backward: 16 step launches: {t_bsteps:7.1f} us ({t_bsteps / S:.2f} us/step);  persistent: {t_bseq:7.1f} us "
      f"({t_bseq / S:.2f} us/step); err {int(sync[1024])}; max |dz diff| {(dZ - dZ2).abs().max().item():.3e}")

"""locally_dense fwd/bwd kernel time for random (synthetic_groups) vs contiguous voxel groups of the same sizes."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import masters_thesis_amd.ops as ops
from masters_thesis_amd.lc_nic import synthetic_groups
be = ops.backend()
B, N, R, D = 64, 20000, 360, 32
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
groups, _ = synthetic_groups(N, R, D, seed=42)
sizes = [len(g) for g in groups]
print("sizes: min %d mean %.0f max %d" % (min(sizes), np.mean(sizes), max(sizes)))
cuts = np.cumsum(sizes)[:-1]
variants = {"random": groups, "contiguous": np.split(np.arange(N), cuts),
            "equal contiguous": np.split(np.arange(N - N % R), R)}
x = torch.randn(B, N, device="cuda")
for name, gs in variants.items():
    goff = torch.tensor(np.concatenate([[0], np.cumsum([len(g) for g in gs])]), dtype=torch.int32, device="cuda")
    idx = torch.tensor(np.concatenate(gs), dtype=torch.int32, device="cuda")
    nW = int(goff[-1])
    W, bias = torch.randn(nW, D, device="cuda") * 0.1, torch.randn(R, D, device="cuda")
    pre, y = torch.zeros(B, R, D, device="cuda"), torch.zeros(B, R, D, device="cuda")
    dpre, dW, db = torch.randn(B, R, D, device="cuda"), torch.zeros(nW, D, device="cuda"), torch.zeros(R, D, device="cuda")
    tf = timeit(lambda: be.locally_dense_fwd(x, N, idx, goff, W, bias, pre, y, B, R, D, 0.2))
    tb = timeit(lambda: be.locally_dense_bwd(x, N, idx, goff, dpre, dW, db, B, R, D))
    print(f"{name:18s}: fwd {tf:6.2f} us  bwd {tb:6.2f} us")
# ---- split mode on the random groups
gs = groups
goffh = np.concatenate([[0], np.cumsum([len(g) for g in gs])])
for piece in (32, 64, 128):
    vg, vr, vf, rf = [0], [], [], [0]
    for r in range(R):
        k = int(goffh[r])
        while True:
            k2 = min(int(goffh[r + 1]), k + piece)
            vg.append(k2); vr.append(r); vf.append(int(k == goffh[r])); k = k2
            if k >= goffh[r + 1]:
                break
        rf.append(len(vr))
    NV = len(vr)
    ti = lambda a: torch.tensor(a, dtype=torch.int32, device="cuda")
    idx = ti(np.concatenate(gs)); vgo, vre, vfi, rfi = ti(vg), ti(vr), ti(vf), ti(rf)
    nW = int(goffh[-1])
    W, bias = torch.randn(nW, D, device="cuda") * 0.1, torch.randn(R, D, device="cuda")
    pre, y, part = torch.zeros(B, R, D, device="cuda"), torch.zeros(B, R, D, device="cuda"), torch.zeros(NV, 64, D, device="cuda")
    dpre, dW, db = torch.randn(B, R, D, device="cuda"), torch.zeros(nW, D, device="cuda"), torch.zeros(R, D, device="cuda")
    tf = timeit(lambda: be.locally_dense_fwd_split(x, N, idx, vgo, vre, rfi, NV, W, bias, pre, y, part, B, R, D, 0.2))
    tb = timeit(lambda: be.locally_dense_bwd_split(x, N, idx, vgo, vre, vfi, NV, dpre, dW, db, B, R, D))
    print(f"split piece={piece:3d} (NV={NV}): fwd {tf:6.2f} us (2 launches)  bwd {tb:6.2f} us")
    xT = x.t().contiguous()
    tf = timeit(lambda: be.locally_dense_fwd_split(xT, B, idx, vgo, vre, rfi, NV, W, bias, pre, y, part, B, R, D, 0.2, voxel_major=True))
    tb = timeit(lambda: be.locally_dense_bwd_split(xT, B, idx, vgo, vre, vfi, NV, dpre, dW, db, B, R, D, voxel_major=True))
    print(f"   voxel-major       : fwd {tf:6.2f} us               bwd {tb:6.2f} us")

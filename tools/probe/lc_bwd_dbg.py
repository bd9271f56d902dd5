"""Debug probe: which elements of dz differ between the backward chain and the per-step kernels (case of tests/test_gpu_ops.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import tests.test_gpu_ops as TT
from masters_thesis_amd import ops
be = ops.backend()
captured = {}
orig = be.lc_seq_bwd
def spy(*a, **k):
    orig(*a, **k)
    torch.cuda.synchronize()
    captured.setdefault("dz", []).append(a[17].clone())
be.lc_seq_bwd = spy
origs = be.lstm_step_bwd
refs = []
def spy2(*a, **k):
    origs(*a, **k)
    refs.append(a[13])
be.lstm_step_bwd = spy2
f = TT.test_lc_seq_bwd_equals_step_kernels
f = getattr(f, "__wrapped__", f)
try:
    f(be, 4, 20, 100, 32, 32, 0.2, 0.3, 0.0)
    print("passed")
except AssertionError as e:
    print("FAILED", str(e)[:120])
got = captured["dz"][0]                     # [T][B][U][4]
T_ = got.shape[0]
ref = torch.stack([r for r in reversed(refs[:T_])])
d = (got - ref).abs()
for i in range(T_):
    di = d[i].amax(dim=-1)                  # [B][U]
    bad = (di > 1e-5)
    print("step", i, "max", float(di.max()), "bad rows", sorted(set(bad.nonzero()[:, 0].tolist()))[:20], "bad unit blocks", sorted(set((bad.nonzero()[:, 1] // 16).tolist()))[:40])

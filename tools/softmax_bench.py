"""Standalone timing of tnt_softmax_cce_f32 at the head shape (960 x 5001, ld 5004)."""
import sys
import torch
sys.path.insert(0, ".")
import os
from masters_thesis_amd import _lib
if os.environ.get("TNT_LIB"): _lib.LIB_PATH = os.environ["TNT_LIB"]
import masters_thesis_amd.ops as ops
be = ops.backend()
rows, V, ld = 960, 5001, 5004
x0 = torch.randn(rows, ld, device="cuda")
tgt = torch.randint(0, V, (rows,), dtype=torch.int32, device="cuda")
loss, corr = torch.zeros(rows, device="cuda"), torch.zeros(rows, device="cuda")
out = torch.zeros(rows, ld, device="cuda")

def timeit(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

x = x0.clone()
print("in place dlogits  : %.2f us" % timeit(lambda: be.softmax_cce(x, tgt, None, loss, corr, x, rows, V, ld, 1e-3)))
print("separate dlogits  : %.2f us" % timeit(lambda: be.softmax_cce(x0, tgt, None, loss, corr, out, rows, V, ld, 1e-3)))
print("probs only        : %.2f us" % timeit(lambda: be.softmax_cce(x0, None, out, None, None, None, rows, V, ld, 0.0)))
print("loss only no write: %.2f us" % timeit(lambda: be.softmax_cce(x0, tgt, None, loss, corr, None, rows, V, ld, 0.0)))
print("copy 19.2 MB      : %.2f us" % timeit(lambda: out.copy_(x0)))

"""Shape fuzz of the attention model's chains against the per-step kernels (the parametrised tests of tests/test_gpu_ops.py
with more shapes: ragged batches, both lane widths, both row-block sizes, tiny and maximal dimensions)."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import tests.test_gpu_ops as TT
from masters_thesis_amd import ops
be = ops.backend()
fw = getattr(TT.test_lc_seq_fwd_equals_step_kernels, "__wrapped__", TT.test_lc_seq_fwd_equals_step_kernels)
bw = getattr(TT.test_lc_seq_bwd_equals_step_kernels, "__wrapped__", TT.test_lc_seq_bwd_equals_step_kernels)
cases = [(1, 1, 1, 4, 4), (2, 9, 33, 8, 12), (3, 64, 384, 32, 32), (2, 65, 130, 32, 32), (2, 128, 192, 64, 64), (3, 17, 192, 36, 64),
         (5, 8, 360, 32, 32), (2, 63, 385, 32, 32), (2, 40, 512, 16, 32), (4, 3, 50, 64, 4)]
bad = 0
for T, B, R, D, A in cases:
    for r_attn, r_in in ((0.2, 0.3), (0.0, 0.0)):
        for name, f, extra in (("fwd", fw, (0,)), ("bwd", bw, (0.0, 0))):
            try:
                f(be, T, B, R, D, A, r_attn, r_in, *extra)
                print("ok  ", name, T, B, R, D, A, r_attn, r_in, flush=True)
            except BaseException as e:
                if type(e).__name__ == "Skipped":
                    print("skip", name, T, B, R, D, A, str(e)[:60]); continue
                bad += 1
                print("FAIL", name, T, B, R, D, A, r_attn, r_in, type(e).__name__, str(e)[:200], flush=True)
print("failures:", bad)

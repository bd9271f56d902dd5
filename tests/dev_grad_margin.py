"""Margins of the full-size gradient check (tests/test_gpu_fullsize.py::test_train_step_matches_oracle_at_full_size): per
variable, the worst err / (1e-4 * max|grad| + 1e-10) over three steps, with model attributes set from the command line
(python tests/dev_grad_margin.py attention fused_bn_drop=0; a development aid, not collected by pytest: it lives under tests/
because it calls the oracle)."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_gpu_fullsize as F
from oracle import models as M

kind = sys.argv[1] if len(sys.argv) > 1 else "attention"
attrs = dict(a.split("=") for a in sys.argv[2:])
rng = np.random.default_rng(21)
rates = tuple(0.0 for _ in F.RATES[kind])
model = F.make(kind, rates=rates)
for k, v in attrs.items():
    setattr(model, k, bool(int(v)))
orc = F.make_oracle(kind, rates)
orc.p = {k: v.astype(np.float64) for k, v in model.get_weights_dict().items()}
names = [k for k in orc.p if "moving_" not in k]
lam = {k: model.arena.entries[k].l2 for k in names}
opt = M.AdamState({k: orc.p[k] for k in names}, lr=1e-4, b1=0.9, b2=0.98, eps=1e-8, clipnorm=0.1)
worst = {}
for step in range(3):
    data, tgt = F.synth(rng)
    F._sync_oracle(model, orc, opt, step)
    w0 = {k: v.copy() for k, v in orc.p.items()}
    res, grads, _ = orc.train_step(data, tgt, opt, M.DropCtx(seed=model.seed, step=step, training=True))
    model.train_step((data, tgt)).as_floats()
    for k in names:
        if grads.get(k) is None:
            continue
        g = model.get_gradient(k).astype(np.float64) + 2 * lam[k] * w0[k]
        scale = np.abs(grads[k]).max()
        r = np.abs(g - grads[k]).max() / (1e-4 * scale + 1e-10)
        worst[k] = max(worst.get(k, 0.0), r)
        if k in ("dense_in/20/bias", "dense_in/0/bias", "dense_in/20/kernel"):
            print(step, k, f"ratio {r:.3f} scale {scale:.3e}")
for k, r in sorted(worst.items(), key=lambda kv: -kv[1])[:8]:
    print(f"{r:8.3f}  {k}")

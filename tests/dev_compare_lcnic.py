import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import models as M
from helpers import synth_batch, tiny_groups
from masters_thesis_amd.lc_nic import NIC
from masters_thesis_amd.optimizers import Adam
dims = (8, 2000, 36, 32, 32, 64, 64, 501, 15)
rates = (0,) * 6
for use_graph in (False, True):
    rng = np.random.default_rng(51)
    B, N, R, D, A, U, Et, V, T = dims
    g = (tiny_groups(N, R, rng), [D] * R)
    model = NIC(g, U, 512, Et, A, V, T, *rates, 0.01, 0.001, 3e-5, 1e-5, seed=11, use_graph=use_graph)
    orc = M.LcNIC(g, U, 512, Et, A, V, T, *rates, 0.01, 0.001, 3e-5, 1e-5).init_params(rng)
    for k, v in orc.p.items():
        model.set_weight(k, v)
    model.compile(Adam(learning_rate=1e-3, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    opt = M.AdamState(orc.p, lr=1e-3, clipnorm=0.1)
    for step in range(4):
        data, tgt = synth_batch(B, N, T, V, U, rng)
        res, grads, _ = orc.train_step(data, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        got = model.train_step((data, tgt)).as_floats()
        ge = model.get_gradient("emb_text/embeddings")
        d = np.abs(ge - grads["emb_text/embeddings"]).max(1)
        bad = np.where(d > 1e-4 * np.abs(grads["emb_text/embeddings"]).max())[0]
        ids = data[1]
        print(f"graph={use_graph} step={step} loss {got['loss']:.6f} vs {res['loss']:.6f} bad rows {bad[:10]} "
              f"counts {[int((ids == b).sum()) for b in bad[:10]]} gmax {np.abs(ge).max():.3e}")
        for k, v in orc.p.items():
            w = model.get_weight(k)
            e = np.abs(w - v).max()
            if e > 2e-5 + 1e-4 * np.abs(v).max() and k != "attention/V/bias":
                print("   weight mismatch", k, e)

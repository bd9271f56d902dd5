"""Generates the frozen known-answer fixtures tests/golden/*.npz (pins 1-5).

    python tests/golden/make_golden.py          # rewrites the .npz files

PARITY UNPINNED: the reference ships no golden vectors and cannot be executed here (TensorFlow is
absent), so these vectors come from this repository's own float64 restatement (oracle/), with
explicitly injected weights and seeded inputs.  They freeze today's behaviour -- of the oracle, of the
Philox dropout stream and of the step semantics -- so that neither the oracle nor the HIP path can
drift silently; they are not outputs of the reference implementation.

Each file holds: the injected weights (``w/<keras name>``), the batches (``x<i>, cap<i>, tgt<i>``),
eval-mode probabilities (+ attention), the step-0 gradients (``g/<name>``), three dropout-free Adam
steps + one step with every dropout rate > 0 (``m<i>/<metric>``, ``p<i>/<name>`` = parameters after
step i) and the greedy decode outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import models as M                      # noqa: E402
from oracle import models_fc as MF                  # noqa: E402
from helpers import synth_batch, tiny_groups        # noqa: E402

SEED, LR = 11, 1e-3
# (B, N, R, D, A, U, Et, V, T): the tiny shape of SURVEY 8c (U, D rounded up to the kernels' tile of 16)
B, N, R, D, A, U, ET, V, T = 3, 37, 4, 16, 3, 16, 6, 11, 4
RATES_LC = (0.1, 0.2, 0.2, 0.2, 0.2, 0.2)


def _run(orc, rates_attr, rates_on, rng, greedy, attention):
    out = {}
    for k, v in orc.p.items():
        out[f"w/{k}"] = v.copy()
    opt = M.AdamState(orc.p, lr=LR, clipnorm=0.1)
    data, tgt = synth_batch(B, N, T, V, U, rng)
    out["x_eval"], out["cap_eval"] = data[0], data[1]
    fwd = orc.forward(data, False)[0]
    if attention:
        out["probs_eval"], out["alpha_eval"] = fwd
    else:
        out["probs_eval"] = fwd
    out.update({f"greedy/{i}": np.asarray(a) for i, a in enumerate(greedy(orc, data[0]))})
    for step in range(4):
        if step == 3:                                  # last step: all dropout sites on (Philox stream pin)
            for a, r in zip(rates_attr, rates_on):
                setattr(orc, a, r)
        data, tgt = synth_batch(B, N, T, V, U, rng, zero_first=(step == 2))
        out[f"x{step}"], out[f"cap{step}"], out[f"tgt{step}"] = data[0], data[1], tgt
        res, grads, _ = orc.train_step(data, tgt, opt, M.DropCtx(seed=SEED, step=step, training=True))
        for k, v in res.items():
            out[f"m{step}/{k}"] = np.float64(v)
        if step == 0:
            for k, g in grads.items():
                if g is not None:
                    out[f"g/{k}"] = g
        for k, v in orc.p.items():
            out[f"p{step}/{k}"] = v.copy()
    return out


def nic_dense():
    rng = np.random.default_rng(101)
    orc = M.NICDense(N, U, ET, V, T, 0.0, 0.0, 0.0, 0.01, 3e-5, 1e-5).init_params(rng)
    z = np.zeros((B, U))
    greedy = lambda o, x: [o.greedy_predict(x, z, z, np.ones(B, np.int64), T)]
    return _run(orc, ("r_in", "r_feat", "r_lstm"), (0.1, 0.2, 0.2), rng, greedy, False)


def lc_nic():
    rng = np.random.default_rng(102)
    g = (tiny_groups(N, R, rng), [D] * R)
    orc = M.LcNIC(g, U, 512, ET, A, V, T, 0, 0, 0, 0, 0, 0, 0.01, 0.001, 3e-5, 1e-5).init_params(rng)
    z = np.zeros((B, U))
    greedy = lambda o, x: o.greedy_predict(x, z, z, np.ones(B, np.int64), T)
    out = _run(orc, ("r_in", "r_feat", "r_text", "r_attn", "r_lstm", "r_out"), RATES_LC, rng, greedy, True)
    out["goff"] = np.concatenate([[0], np.cumsum([len(i) for i in g[0]])]).astype(np.int64)
    out["gidx"] = np.concatenate(g[0]).astype(np.int64)
    return out


def fc_nic():
    rng = np.random.default_rng(103)
    orc = MF.FcNIC(N, U, ET, ET, V, T, 0, 0, 0, 0, 0, 0.01, 3e-5, 1e-5).init_params(rng)
    z = np.zeros((B, U))
    greedy = lambda o, x: [o.greedy_predict(x, z, z, np.ones(B, np.int64), T)]
    return _run(orc, ("r_in", "r_feat", "r_text", "r_lstm", "r_out"), (0.1, 0.2, 0.1, 0.2, 0.3), rng, greedy, False)


def ms2_tiny():
    """ms2_NIC (two subjects, shared decoder): three train steps (its sub-calls always run with training=True, the
    feature Dropout is applied twice; all rates on) + one test_step."""
    from oracle.models_ms import MsLcNIC
    rng = np.random.default_rng(104)
    S, Bs = 2, 2
    g = (tiny_groups(N, R, rng), [D] * R)
    orc = MsLcNIC(g, U, 512, ET, A, V, T, *RATES_LC, 0.01, 0.001, 3e-5, 1e-5, n_subjects=S).init_params(rng)
    out = {f"w/{k}": v.copy() for k, v in orc.p.items()}
    out["goff"] = np.concatenate([[0], np.cumsum([len(i) for i in g[0]])]).astype(np.int64)
    out["gidx"] = np.concatenate(g[0]).astype(np.int64)
    opt = M.AdamState(orc.p, lr=LR, clipnorm=0.1)
    for step in range(3):
        data, tgt = synth_batch(S * Bs, N, T, V, U, rng)
        out[f"x{step}"], out[f"cap{step}"], out[f"tgt{step}"] = data[0], data[1], tgt
        res, _, _ = orc.train_step(data, tgt, opt, M.DropCtx(seed=SEED, step=step, training=True))
        for k, v in res.items():
            out[f"m{step}/{k}"] = np.float64(v)
        for k, v in orc.p.items():
            out[f"p{step}/{k}"] = v.copy()
    data, tgt = synth_batch(S * Bs, N, T, V, U, rng)
    out["x_test"], out["cap_test"], out["tgt_test"] = data[0], data[1], tgt
    res, _ = orc.test_step(data, tgt, M.DropCtx(seed=SEED, step=3, training=True))
    for k, v in res.items():
        out[f"mtest/{k}"] = np.float64(v)
    return out


MID = dict(B=8, N=2000, R=36, D=32, A=32, U=64, ET=64, V=501, T=15)       # the mid-size shape of SURVEY 8c(2)


def lc_nic_mid():
    """call_attention at the mid-size shape: eval probabilities, attention maps and the greedy caption ids.
    Stored as float32 (weights are rounded to float32 BEFORE the oracle runs, so the file is self-consistent)."""
    m = MID
    rng = np.random.default_rng(105)
    g = (tiny_groups(m["N"], m["R"], rng), [m["D"]] * m["R"])
    orc = M.LcNIC(g, m["U"], 512, m["ET"], m["A"], m["V"], m["T"], 0, 0, 0, 0, 0, 0, 0.01, 0.001, 3e-5, 1e-5).init_params(rng)
    orc.p = {k: v.astype(np.float32).astype(np.float64) for k, v in orc.p.items()}
    out = {f"w/{k}": v.astype(np.float32) for k, v in orc.p.items()}
    out["goff"] = np.concatenate([[0], np.cumsum([len(i) for i in g[0]])]).astype(np.int64)
    out["gidx"] = np.concatenate(g[0]).astype(np.int64)
    data, _ = synth_batch(m["B"], m["N"], m["T"], m["V"], m["U"], rng)
    out["x_eval"], out["cap_eval"] = data[0], data[1]
    (probs, alpha), _ = orc.forward(data, False)
    out["probs_eval"], out["alpha_eval"] = probs.astype(np.float32), alpha.astype(np.float32)
    z = np.zeros((m["B"], m["U"]))
    words, gp, _, _ = orc.greedy_predict(data[0], z, z, np.ones(m["B"], np.int64), m["T"])
    out["greedy_words"], out["greedy_probs"] = words, gp.astype(np.float32)
    return out


FIXTURES = {"nic_dense_tiny": nic_dense, "lc_nic_tiny": lc_nic, "fc_nic_tiny": fc_nic, "ms2_tiny": ms2_tiny,
            "lc_nic_mid": lc_nic_mid}


def closed_forms():
    """Loss closed forms of SURVEY 8c(5): the scratch check of AttemptFour/temp.py:73-90
    (CE of p=0.6 and p=0.93 targets), the clip path p < 1e-7, and argmax tie-break = first maximum."""
    return {"ce_0.6": -np.log(0.6), "ce_0.93": -np.log(0.93), "ce_clip": -np.log(1e-7)}


if __name__ == "__main__":
    for name, fn in FIXTURES.items():
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **fn())
        print(path, os.path.getsize(path), "bytes")

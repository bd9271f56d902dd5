"""Frozen known-answer vectors (tests/golden/*.npz, written by tests/golden/make_golden.py).

CPU part: the oracle reproduces every stored array (guards the checker itself against drift).
GPU part: the HIP path, loaded with the stored weights, reproduces the stored probabilities,
greedy captions, metrics and post-Adam parameters -- three dropout-free steps and one step with every
dropout site switched on (pins the Philox stream wiring).  PARITY UNPINNED, see make_golden.py."""
import os
import sys

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import make_golden as G          # noqa: E402


def load(name):
    with np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", sorted(G.FIXTURES))
def test_oracle_reproduces_golden(name):
    want, got = load(name), G.FIXTURES[name]()
    assert set(want) == set(got)
    for k, v in want.items():
        g = np.asarray(got[k])
        assert g.shape == v.shape, k
        if v.dtype.kind in "iub":
            assert np.array_equal(g, v), k
        else:
            assert np.allclose(g, v, rtol=1e-10, atol=1e-13), (k, np.abs(g - v).max())


def test_closed_forms():
    from oracle import ops as O
    cf = G.closed_forms()
    p = np.array([[0.05, 0.6, 0.35], [0.93, 0.03, 0.04]])
    ce = O.cce_from_probs(p, np.array([1, 0]))
    assert np.allclose(ce, [cf["ce_0.6"], cf["ce_0.93"]], rtol=1e-12)           # temp.py:73-90
    tiny = np.array([[1e-12, 1.0 - 1e-12]])
    assert np.allclose(O.cce_from_probs(tiny, np.array([0])), cf["ce_clip"], rtol=1e-12)
    assert O.accuracy(np.array([[0.4, 0.4, 0.2]]), np.array([0])) == 1.0        # tie -> first maximum


# ------------------------------------------------------------------------------------- GPU
def _weights(gold, prefix):
    return {k[len(prefix):]: v for k, v in gold.items() if k.startswith(prefix)}


def _check_steps(model, gold, rates_attr, rates_on, skip=()):
    from masters_thesis_amd.optimizers import Adam
    B, U = G.B, G.U
    saved = {a: getattr(model, a) for a in rates_attr}
    for a in rates_attr:
        setattr(model, a, 0.0)
    model.compile(Adam(learning_rate=G.LR, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    z = np.zeros((B, U), np.float32)
    for step in range(4):
        if step == 3:
            for a in rates_attr:
                setattr(model, a, saved[a])
            model._graphs = {}
        data = (gold[f"x{step}"], gold[f"cap{step}"], z, z)
        got = model.train_step((data, gold[f"tgt{step}"])).as_floats()
        for k in got:
            if k == "lr":
                continue
            want = float(gold[f"m{step}/{k}"])
            assert abs(got[k] - want) <= 1e-4 * abs(want) + 1e-6, (step, k, got[k], want)
        for k, v in _weights(gold, f"p{step}/").items():
            if k in skip:
                continue
            w = model.get_weight(k)
            tol = 2e-2 * G.LR + 1e-4 * np.abs(v).max()
            if f"g/{k}" in gold:      # zero-gradient elements: float32 rounding noise through Adam's g/(|g|+eps)
                tol = tol + G.LR * (step + 1) * (np.abs(gold[f"g/{k}"]) < 1e-8)
            assert (np.abs(w - v) <= tol).all(), (step, k, np.abs(w - v).max())


@pytest.mark.gpu
def test_gpu_nic_dense_golden():
    from masters_thesis_amd.nic import NIC
    gold = load("nic_dense_tiny")
    model = NIC(G.N, G.U, G.ET, G.V, G.T, 0.1, 0.2, 0.2, 0.01, 3e-5, 1e-5, seed=G.SEED)
    for k, v in _weights(gold, "w/").items():
        model.set_weight(k, v)
    z = np.zeros((G.B, G.U), np.float32)
    p = model((gold["x_eval"], gold["cap_eval"], z, z)).cpu().numpy()
    assert np.abs(p - gold["probs_eval"]).max() <= 1e-5
    gp = model.greedy_predict(gold["x_eval"], z, z, np.ones(G.B, np.int64), G.T, G.U)
    assert np.array_equal(gp.argmax(-1), gold["greedy/0"].argmax(-1)) and np.abs(gp - gold["greedy/0"]).max() <= 1e-5
    _check_steps(model, gold, ("r_in", "r_feat", "r_lstm"), (0.1, 0.2, 0.2))


@pytest.mark.gpu
def test_gpu_lc_nic_golden():
    from masters_thesis_amd.lc_nic import NIC
    gold = load("lc_nic_tiny")
    groups = [gold["gidx"][a:b] for a, b in zip(gold["goff"][:-1], gold["goff"][1:])]
    model = NIC((groups, [G.D] * G.R), G.U, 512, G.ET, G.A, G.V, G.T, *G.RATES_LC, 0.01, 0.001, 3e-5, 1e-5, seed=G.SEED)
    for k, v in _weights(gold, "w/").items():
        model.set_weight(k, v)
    z = np.zeros((G.B, G.U), np.float32)
    p, alpha = model((gold["x_eval"], gold["cap_eval"], z, z))
    assert np.abs(p.cpu().numpy() - gold["probs_eval"]).max() <= 1e-5
    assert np.abs(alpha.cpu().numpy() - gold["alpha_eval"]).max() <= 1e-5
    words, probs, al, s = model.greedy_predict(gold["x_eval"], z, z, np.ones(G.B, np.int64), G.T, G.U, None)
    assert np.array_equal(words, gold["greedy/0"])
    assert np.abs(probs - gold["greedy/1"]).max() <= 1e-5 and np.abs(al - gold["greedy/2"]).max() <= 1e-5
    assert np.abs(s - gold["greedy/3"]).max() <= 1e-5
    _check_steps(model, gold, ("r_in", "r_feat", "r_text", "r_attn", "r_lstm", "r_out"), G.RATES_LC,
                 skip=("attention/V/bias",))


@pytest.mark.gpu
def test_gpu_fc_nic_golden():
    from masters_thesis_amd.fc_nic import NICfc
    gold = load("fc_nic_tiny")
    model = NICfc(G.N, G.U, G.ET, G.ET, G.V, G.T, 0.1, 0.2, 0.1, 0.2, 0.3, 0.01, 3e-5, 1e-5, seed=G.SEED)
    for k, v in _weights(gold, "w/").items():
        model.set_weight(k, v)
    z = np.zeros((G.B, G.U), np.float32)
    p, _ = model((gold["x_eval"], gold["cap_eval"], z, z))
    assert np.abs(p.cpu().numpy() - gold["probs_eval"]).max() <= 1e-5
    ids = model.greedy_predict(gold["x_eval"], z, z, np.ones(G.B, np.int64), G.T)
    assert np.array_equal(ids, gold["greedy/0"])
    _check_steps(model, gold, ("r_in", "r_feat", "r_text", "r_lstm", "r_out"), (0.1, 0.2, 0.1, 0.2, 0.3))


@pytest.mark.gpu
def test_gpu_ms2_golden():
    """ms2_NIC (two subjects) on the HIP path against the frozen vectors: three train steps + one test_step."""
    from masters_thesis_amd.ms_nic import NIC
    from masters_thesis_amd.optimizers import Adam
    gold = load("ms2_tiny")
    groups = [gold["gidx"][a:b] for a, b in zip(gold["goff"][:-1], gold["goff"][1:])]
    model = NIC((groups, [G.D] * G.R), G.U, 512, G.ET, G.A, G.V, G.T, *G.RATES_LC, 0.01, 0.001, 3e-5, 1e-5, n_subjects=2,
                seed=G.SEED)
    for k, v in _weights(gold, "w/").items():
        model.set_weight(k, v)
    model.compile(Adam(learning_rate=G.LR, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    B = gold["x0"].shape[0]
    z = np.zeros((B, G.U), np.float32)
    for step in range(3):
        got = model.train_step(((gold[f"x{step}"], gold[f"cap{step}"], z, z), gold[f"tgt{step}"])).as_floats()
        for k in got:
            if k == "lr":
                continue
            want = float(gold[f"m{step}/{k}"])
            assert abs(got[k] - want) <= 1e-4 * abs(want) + 1e-6, (step, k, got[k], want)
        for k, v in _weights(gold, f"p{step}/").items():
            if k == "attention/V/bias":
                continue
            assert np.abs(model.get_weight(k) - v).max() <= 2e-2 * G.LR + 1e-4 * np.abs(v).max(), (step, k)
    got = model.test_step(((gold["x_test"], gold["cap_test"], z, z), gold["tgt_test"])).as_floats()
    for k in got:
        want = float(gold[f"mtest/{k}"])
        assert abs(got[k] - want) <= 1e-4 * abs(want) + 1e-6, (k, got[k], want)


@pytest.mark.gpu
def test_gpu_lc_nic_mid_golden():
    """call_attention at the mid-size shape (B=8, N=2000, R=36, U=64, V=501, T=15): eval probabilities, attention maps
    and greedy captions against the frozen vectors."""
    from masters_thesis_amd.lc_nic import NIC
    gold = load("lc_nic_mid")
    m = G.MID
    groups = [gold["gidx"][a:b] for a, b in zip(gold["goff"][:-1], gold["goff"][1:])]
    model = NIC((groups, [m["D"]] * m["R"]), m["U"], 512, m["ET"], m["A"], m["V"], m["T"], 0, 0, 0, 0, 0, 0, 0.01, 0.001,
                3e-5, 1e-5, seed=G.SEED)
    for k, v in _weights(gold, "w/").items():
        model.set_weight(k, v)
    z = np.zeros((m["B"], m["U"]), np.float32)
    p, alpha = model((gold["x_eval"], gold["cap_eval"], z, z))
    p = p.cpu().numpy()
    assert np.abs(np.log(p) - np.log(gold["probs_eval"])).max() <= 1e-4 * max(1.0, np.abs(np.log(gold["probs_eval"])).max())
    assert np.abs(alpha.cpu().numpy() - gold["alpha_eval"]).max() <= 1e-5
    words, probs, _, _ = model.greedy_predict(gold["x_eval"], z, z, np.ones(m["B"], np.int64), m["T"], m["U"], None)
    assert np.array_equal(words, gold["greedy_words"])
    assert np.abs(probs - gold["greedy_probs"]).max() <= 1e-4

"""GPU parity of the GRU step kernels and of the ThinkAndTell/att_model.py generator against the float64 oracle."""
import numpy as np
import pytest
import torch

from oracle import models as M
from oracle import ops as O
from oracle.models_att import CaptionGeneratorAtt

pytestmark = pytest.mark.gpu


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


def il3(w, U):
    from masters_thesis_amd.think_and_tell_att import interleave3
    return interleave3(np.asarray(w, np.float32), U)


def close(got, want, rtol=1e-4):
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert np.abs(got - want).max() <= rtol * (np.abs(want).max() + 1e-30), np.abs(got - want).max()


@pytest.mark.parametrize("B,U", [(3, 16), (64, 512), (20, 80)])
def test_gru_step_kernels(B, U):
    import masters_thesis_amd.ops as ops
    from masters_thesis_amd.think_and_tell_att import deinterleave3
    be = ops.backend()
    rng = np.random.default_rng(41)
    xz = rng.standard_normal((B, 3 * U))
    h = rng.standard_normal((B, U)) * 0.5
    Uk = rng.standard_normal((U, 3 * U)) / np.sqrt(U)
    br = 0.1 * rng.standard_normal(3 * U)
    h2, cache = O.gru_step_fwd(xz, h, Uk, br)
    f = lambda *s: torch.zeros(*s, dtype=torch.float32, device="cuda")
    hd, gates = f(B, U), f(B, U, 4)
    be.gru_step_fwd(dev(il3(xz, U)), dev(h), dev(il3(Uk, U)), dev(il3(br, U)), hd, gates, B, U)
    close(hd, h2)
    z, r, hh, rech, _ = cache
    close(gates, np.stack([z, r, hh, rech], axis=-1))
    # backward: with and without the recurrent matmul of the following step
    dh_ext, dh_pass = rng.standard_normal((B, U)), rng.standard_normal((B, U))
    drec_next = rng.standard_normal((B, 3 * U)) * 0.3
    for use_next in (False, True):
        dh = dh_ext + dh_pass + (drec_next @ Uk.T if use_next else 0.0)
        dxz_w, drec_w, _ = O.gru_step_bwd(dh, cache, Uk)
        dxz, drec, dpo = f(B, U, 4), f(B, U, 4), dev(dh_pass)
        be.gru_step_bwd(dev(il3(drec_next, U)) if use_next else None, dev(il3(Uk, U)), dpo, dev(dh_ext), gates, dev(h), dxz,
                        drec, dpo, B, U)
        close(deinterleave3(dxz.cpu().numpy()), dxz_w)
        close(deinterleave3(drec.cpu().numpy()), drec_w)
        close(dpo, dh * z)
        assert float(dxz[..., 3].abs().max()) == 0.0 and float(drec[..., 3].abs().max()) == 0.0


def batch(rng, B, N, T, V):
    x = rng.standard_normal((B, N)).astype(np.float32)
    tgt = rng.integers(1, V, (B, T)).astype(np.int32)
    for b in range(B):
        tgt[b, rng.integers(2, T + 1):] = 0
    return x, tgt


@pytest.mark.parametrize("dims", [(3, 37, 8, 16, 11, 4), (8, 2000, 64, 64, 501, 15)])
@pytest.mark.parametrize("drop", [0.0, 0.3])
def test_att_model_train_parity(dims, drop):
    from masters_thesis_amd import think_and_tell_att as ATT
    from masters_thesis_amd.optimizers import Adam
    B, N, E, U, V, T = dims
    rng = np.random.default_rng(83)
    orc = CaptionGeneratorAtt(N, E, U, V, T, l2_reg=0.01, dropout=drop).init_params(rng)
    model = ATT.CaptionGenerator(ATT.Encoder(E, 0.01, "glorot_uniform", drop), ATT.Decoder(E, U, V, 0.01, "glorot_uniform", drop),
                                 None, T, seed=11)
    model.compile(Adam(learning_rate=1e-3, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    opt = M.AdamState(orc.p, lr=1e-3, clipnorm=0.1)
    for step in range(4):                       # eager, capture, replay, replay
        x, tgt = batch(rng, B, N, T, V)
        if step == 0:
            model._stage(x, tgt)
            for k, v in orc.p.items():
                model.set_weight(k, v)
        res, grads = orc.train_step(x, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        got = model.train_step((x, None, tgt)).as_floats()
        for k in res:
            assert abs(got[k] - res[k]) <= 1e-4 * abs(res[k]) + 1e-7, (step, k, got[k], res[k])
        bad = 0
        for k, v in orc.p.items():
            w = model.get_weight(k)
            tol = 2e-2 * 1e-3 + 1e-4 * np.abs(v).max()
            # relu gates of fc1 / fc2 / encoder can flip on float32 rounding at the big shape
            bad += int((np.abs(w - v) > tol).sum())
        total = sum(v.size for v in orc.p.values())
        assert bad <= 2e-4 * total, (step, bad, total)
    x, tgt = batch(rng, B, N, T, V)
    res = orc.test_step(x, tgt)
    got = model.test_step((x, None, tgt)).as_floats()
    for k in res:
        assert abs(got[k] - res[k]) <= 1e-4 * abs(res[k]) + 1e-7, k

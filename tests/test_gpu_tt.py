"""ThinkAndTell / ShowAndTell caption generators on the GPU against the float64 oracle."""
import numpy as np
import pytest

from oracle import models as M
from oracle.models_tt import CaptionGeneratorTT

pytestmark = pytest.mark.gpu


def batch(rng, B, N, T, V):
    x = rng.standard_normal((B, N)).astype(np.float32)
    tgt = rng.integers(1, V, (B, T)).astype(np.int32)
    for b in range(B):
        tgt[b, rng.integers(2, T + 1):] = 0
    return x, tgt


# the fourth case is BASELINE config 1 at its stated size: ShowAndTell/model.py:40-65,125-164 on 4096-d image features
# (train.py:165-225) with long, padded captions (T = 40, masked tail, loss over i = 1..T-1)
@pytest.mark.parametrize("sat,drop,dims", [(False, 0.3, (8, 500, 64, 64, 301, 12)), (True, 0.0, (6, 256, 32, 32, 101, 9)),
                                           (False, 0.0, (64, 5000, 512, 512, 5001, 15)),
                                           (True, 0.0, (32, 4096, 512, 512, 5001, 40))])
def test_tt_parity(sat, drop, dims):
    from masters_thesis_amd import think_and_tell as TT, show_and_tell as SAT
    from masters_thesis_amd.optimizers import Adam
    rng = np.random.default_rng(91)
    B, N, E, U, V, T = dims
    orc = CaptionGeneratorTT(N, E, U, V, T, l2_reg=0.001, dropout=drop, show_and_tell=sat).init_params(rng)
    if sat:
        model = SAT.CaptionGenerator(SAT.Encoder(E), SAT.Decoder(E, U, V), None, T, seed=11)
    else:
        model = TT.CaptionGenerator(TT.Encoder(E, 0.001, "glorot_uniform", drop),
                                    TT.Decoder(E, U, V, 0.001, "glorot_uniform", drop), None, T, seed=11)
    model.compile(Adam(learning_rate=1e-3, beta_2=0.98, epsilon=1e-8))
    opt = M.AdamState(orc.p, lr=1e-3, clipnorm=None)
    big = B * N > 100000
    for step in range(2 if big else 4):
        x, tgt = batch(rng, B, N, T, V)
        if step == 0:
            model._stage(x, tgt)
            for k, v in orc.p.items():
                model.set_weight(k, v)
        res, _ = orc.train_step(x, tgt, opt, M.DropCtx(seed=11, step=step, training=True))
        got = model.train_step((x, tgt) if sat else (x, None, tgt)).as_floats()
        for k in res:
            assert abs(got[k] - res[k]) <= 1e-4 * abs(res[k]) + 1e-6, (step, k, got[k], res[k])
        for k, v in orc.p.items():
            w = model.get_weight(k)
            bad = np.abs(w - v) > 2e-2 * 1e-3 + 1e-4 * np.abs(v).max()
            # the head is Dense(V, relu): a pre-activation within float32 rounding of 0 can flip its relu
            # gate between the float32 kernel and the float64 oracle, which moves one kernel column by
            # O(lr) under Adam -- a measure-zero discontinuity, not an arithmetic error.  Allow it at the
            # full-size case only, and only for a vanishing fraction of elements.
            assert bad.mean() <= (2e-4 if big else 0.0), (step, k, bad.mean(), np.abs(w - v).max())

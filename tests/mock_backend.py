"""CPU stand-in for the kernel backend -- TEST INFRASTRUCTURE ONLY.

Implements the method surface of ``masters_thesis_amd.ops.HipBackend`` on CPU torch tensors
with numpy (float32 storage, float64 arithmetic inside each op), following the semantics
documented in include/tnt_hip.h and the oracle ops.  It exists so that the *host
orchestration* (launch order, buffer wiring, layouts, optimizer plumbing) can be checked
against the model-level oracle without a GPU.  The product never imports this file; the
HIP kernels themselves are checked on the GPU by tests/test_gpu_ops.py.
"""
import numpy as np
import torch

from oracle import ops as O
from oracle.philox import uniform24


def flat(t):
    """1-D numpy view of the storage from t's first element to the end of its storage."""
    if t is None:
        return None
    n = t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()
    return torch.empty(0, dtype=t.dtype).set_(t.untyped_storage(), t.storage_offset(), (n,), (1,)).numpy()


def mat(t, rows, cols, ld):
    return np.lib.stride_tricks.as_strided(flat(t), (rows, cols), (ld * 4, 4))


def _keep(e, rate, seed, site, step):
    """keep decision for logical element indices e (int64 array)."""
    e = np.asarray(e, dtype=np.uint64)
    shp = e.shape
    e = e.reshape(-1)
    from oracle.philox import philox4x32_10
    grp = e >> np.uint64(2)
    r = philox4x32_10((grp & np.uint64(0xFFFFFFFF)).astype(np.uint32), (grp >> np.uint64(32)).astype(np.uint32),
                      np.uint32(site), np.uint32(step), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    r = np.stack(r, -1)[np.arange(e.size), (e & np.uint64(3)).astype(np.int64)]
    u = (r >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (u >= np.float32(rate)).reshape(shp)


class MockBackend:
    name = "mock-cpu"

    def bn_nchunk(self, rows):
        return max(2, (rows + 63) // 64)        # >= 2: the mock's synchronised-BatchNorm partials need 3 * C floats

    # ---------------------------------------------------------------- gemm
    def gemm(self, A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, pre=None, act=0,
             slope=0.2, accumulate=False, splitk=1, work=None):
        a = mat(A, K, M, lda).T if transA else mat(A, M, K, lda)
        b = mat(B, N, K, ldb).T if transB else mat(B, K, N, ldb)
        v = a.astype(np.float64) @ b.astype(np.float64)
        if bias is not None:
            v = v + flat(bias)[:N]
        if pre is not None:
            mat(pre, M, N, ldc)[...] = v
        v = O.act_fwd(v, act, slope)
        c = mat(C, M, N, ldc)
        c[...] = (c + v) if accumulate else v

    def dropout(self, x, y, rows, cols, ld, tmajor_B, lwidth, lcol0, rate, seed, site, step, step_dev=None,
                rows_per_site=0):
        if step_dev is not None:
            step = (step + int(step_dev[0])) & 0xFFFFFFFF
        if rows_per_site > 0:
            assert tmajor_B == 0
            for blk in range(0, rows, rows_per_site):
                nb = min(rows_per_site, rows - blk)
                self.dropout(x.view(-1)[blk * ld:], y.view(-1)[blk * ld:], nb, cols, ld, 0, lwidth, lcol0, rate, seed,
                             site + blk // rows_per_site, step)
            return
        xs, ys = mat(x, rows, cols, ld), mat(y, rows, cols, ld)
        r = np.arange(rows)
        T = rows // tmajor_B if tmajor_B > 0 else 0
        lrow = (r % tmajor_B) * T + r // tmajor_B if tmajor_B > 0 else r
        e = lrow[:, None].astype(np.int64) * lwidth + lcol0 + np.arange(cols)[None, :]
        k = _keep(e, rate, seed, site, step)
        scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
        ys[...] = np.where(k, xs * scale, np.float32(0))

    def dropout2(self, x, y, rows, cols, ld, mask_a, mask_b, seed, step, step_dev=None):
        for k, (tB, lw, lc0, rps, rate, site) in enumerate((mask_a, mask_b)):
            self.dropout(x if k == 0 else y, y, rows, cols, ld, tB, lw, lc0, rate, seed, site, step, step_dev, rows_per_site=rps)

    def dropout_mask4(self, out, n, nsites, rate, seed, site0, step, step_dev=None):
        if step_dev is not None:
            step = (step + int(step_dev[0])) & 0xFFFFFFFF
        o = out.view(-1).numpy()
        for k in range(nsites):
            bits = _keep(np.arange(n), rate, seed, site0 + k, step).reshape(n // 4, 4)
            o[k * (n // 4):(k + 1) * (n // 4)] = (bits * np.array([1, 2, 4, 8])).sum(1).astype(np.uint8)

    @staticmethod
    def _check_keep4(keep4, keep, n):
        """The stored bits handed to an attention step must be the Philox mask that step would have generated."""
        if keep4 is None or keep is None:
            return
        by = keep4.view(-1).numpy()[:n // 4]
        bits = ((by[:, None] >> np.arange(4)[None, :]) & 1).astype(bool).reshape(-1)
        assert np.array_equal(bits, keep.reshape(-1)), "keep4 does not match the step's dropout stream"

    def act_bwd(self, pre, dy, dx, n, act, slope=0.2):
        flat(dx)[:n] = O.act_bwd(flat(pre)[:n].astype(np.float64), flat(dy)[:n].astype(np.float64), act, slope)

    # ---------------------------------------------------------------- norms
    def batchnorm_fwd(self, x, gamma, beta, mov_mean, mov_var, y, xhat, inv_std, rows, C, ldy, training, eps,
                      momentum, work, drop=None):
        xs = mat(x, rows, C, C).astype(np.float64)
        mm, mv = flat(mov_mean)[:C], flat(mov_var)[:C]
        yo, (xh, inv, _), nmm, nmv = O.batchnorm_fwd(xs, flat(gamma)[:C].astype(np.float64),
                                                     flat(beta)[:C].astype(np.float64), mm.astype(np.float64),
                                                     mv.astype(np.float64), bool(training), eps, momentum)
        mat(y, rows, C, ldy)[...] = yo
        mat(xhat, rows, C, C)[...] = xh
        flat(inv_std)[:C] = inv
        mm[...] = nmm
        mv[...] = nmv
        if drop is not None and drop[0] > 0:
            rate, seed, site, step_dev = drop
            self.dropout(y, y, rows, C, ldy, 0, C, 0, rate, seed, site, 0, step_dev)

    def batchnorm_bwd(self, dy, xhat, gamma, inv_std, dx, dgamma, dbeta, rows, C, lddy, training, work, act_pre=None,
                      slope=0.0):
        cache = (mat(xhat, rows, C, C).astype(np.float64), flat(inv_std)[:C].astype(np.float64), bool(training))
        dxo, dg, db = O.batchnorm_bwd(mat(dy, rows, C, lddy).astype(np.float64), flat(gamma)[:C].astype(np.float64), cache)
        if dx is not None:
            if act_pre is not None:
                dxo = dxo * np.where(mat(act_pre, rows, C, C) > 0, 1.0, slope)
            mat(dx, rows, C, C)[...] = dxo
        if dgamma is not None:
            flat(dgamma)[:C] = dg
            flat(dbeta)[:C] = db

    # synchronised BatchNorm pieces: the mock's "partials" are per-column (n, sum, sum of squares) in float64-exact float32s
    def batchnorm_stats(self, x, rows, C, part):
        xs = mat(x, rows, C, C).astype(np.float64)
        p = flat(part)
        p[:self.bn_nchunk(rows) * 2 * C] = 0
        p[0:C] = rows; p[C:2 * C] = xs.sum(0); p[2 * C:3 * C] = (xs * xs).sum(0)

    def batchnorm_apply_stats(self, part_all, nrep, x, gamma, beta, mov_mean, mov_var, y, xhat, inv_std, rows, C, ldy, eps,
                              momentum, mean_work):
        stride = self.bn_nchunk(rows) * 2 * C
        p = flat(part_all)[:nrep * stride].reshape(nrep, stride)[:, :3 * C].reshape(nrep, 3, C).astype(np.float64)
        n, s1, s2 = p[:, 0].sum(0), p[:, 1].sum(0), p[:, 2].sum(0)
        mean = s1 / n
        var = s2 / n - mean * mean
        inv = 1.0 / np.sqrt(var + eps)
        xs = mat(x, rows, C, C).astype(np.float64)
        xh = (xs - mean) * inv
        mat(xhat, rows, C, C)[...] = xh
        mat(y, rows, C, ldy)[...] = xh * flat(gamma)[:C] + flat(beta)[:C]
        flat(inv_std)[:C] = inv
        mm, mv = flat(mov_mean)[:C], flat(mov_var)[:C]
        mm[...] = mm * momentum + mean * (1 - momentum)
        mv[...] = mv * momentum + var * (1 - momentum)

    def batchnorm_dx(self, dy, lddy, xhat, gamma, inv_std, dgamma_sum, dbeta_sum, dx, rows, C, n_total):
        g = mat(dy, rows, C, lddy).astype(np.float64)
        xh = mat(xhat, rows, C, C).astype(np.float64)
        k = flat(gamma)[:C].astype(np.float64) * flat(inv_std)[:C]
        mat(dx, rows, C, C)[...] = k / n_total * (n_total * g - flat(dbeta_sum)[:C] - xh * flat(dgamma_sum)[:C])

    def ln_lstm_cell_fwd(self, zk, zr, bias, c_prev, gamma_s, beta_s, gates, chat, istd, c, h, B, U, eps):
        """include/tnt_hip.h: tnt_ln_lstm_cell_fwd_f32 (gate-interleaved [B][U][4] tensors)"""
        z = (flat(zk)[:B * U * 4].astype(np.float64) + flat(zr)[:B * U * 4]).reshape(B, U, 4) + flat(bias)[:U * 4].astype(np.float64).reshape(1, U, 4)
        gi, gf, gg, go = O.sigmoid(z[..., 0]), O.sigmoid(z[..., 1]), np.tanh(z[..., 2]), O.sigmoid(z[..., 3])
        craw = gf * flat(c_prev)[:B * U].astype(np.float64).reshape(B, U) + gi * gg
        cn, (xh, inv) = O.layernorm_fwd(craw, flat(gamma_s)[:U].astype(np.float64), flat(beta_s)[:U].astype(np.float64), eps)
        flat(gates)[:B * U * 4] = np.stack([gi, gf, gg, go], -1).reshape(-1)
        flat(chat)[:B * U] = xh.reshape(-1); flat(istd)[:B] = inv.reshape(-1)
        flat(c)[:B * U] = cn.reshape(-1); flat(h)[:B * U] = (go * np.tanh(cn)).reshape(-1)

    def ln_lstm_cell_bwd(self, dh_a, dh_b, dh_c, dcn_in, gates, c_prev, c, chat, istd, gamma_s, dz, dc_prev, dcnt, B, U):
        f64 = lambda t, *sh: flat(t)[:int(np.prod(sh))].astype(np.float64).reshape(*sh)
        dh = sum(f64(t, B, U) for t in (dh_a, dh_b, dh_c) if t is not None)
        g4 = f64(gates, B, U, 4)
        gi, gf, gg, go = g4[..., 0], g4[..., 1], g4[..., 2], g4[..., 3]
        tc = np.tanh(f64(c, B, U))
        dcn = (f64(dcn_in, B, U) if dcn_in is not None else 0) + dh * go * (1 - tc * tc)
        dcr, _, _ = O.layernorm_bwd(dcn, f64(gamma_s, U), (f64(chat, B, U), f64(istd, B)[:, None]))
        cp = f64(c_prev, B, U)
        out = np.stack([dcr * gg * gi * (1 - gi), dcr * cp * gf * (1 - gf), dcr * gi * (1 - gg * gg), dh * tc * go * (1 - go)], -1)
        flat(dcnt)[:B * U] = dcn.reshape(-1)
        flat(dz)[:B * U * 4] = out.reshape(-1)
        flat(dc_prev)[:B * U] = (dcr * gf).reshape(-1)

    def enc_tail_fwd(self, y, gamma, beta, mov_mean, mov_var, out, xhat, inv_std, rows, C, ldo, training, eps, momentum,
                     r_feat, r_lstm, seed, site_feat, site_lstm, step_dev=None):
        """composition the fused kernel stands for: dropout -> batchnorm -> dropout"""
        tmp = torch.zeros(rows, C, dtype=torch.float32)
        tmp.copy_(y.view(-1)[:rows * C].view(rows, C))
        if training and r_feat > 0:
            self.dropout(tmp, tmp, rows, C, C, 0, C, 0, r_feat, seed, site_feat, 0, step_dev)
        self.batchnorm_fwd(tmp, gamma, beta, mov_mean, mov_var, out, xhat, inv_std, rows, C, ldo, training, eps,
                           momentum, None)
        if training and r_lstm > 0:
            self.dropout(out, out, rows, C, ldo, 0, C, 0, r_lstm, seed, site_lstm, 0, step_dev)

    def dense_fwd_stream(self, x, w, part, B, E, K, ldx, ldw, nsplit):
        """K-split partials in the kernel's tile order: 16-k tile i belongs to split (i % (4 nsplit)) // 4"""
        X = mat(x, B, K, ldx).astype(np.float64)
        W = mat(w, K, E, ldw).astype(np.float64)
        out = flat(part)[:nsplit * B * E].reshape(nsplit, B, E)
        out[...] = 0
        for i in range((K + 15) // 16):
            sp = (i % (4 * nsplit)) // 4
            out[sp] += X[:, 16 * i:16 * i + 16] @ W[16 * i:16 * i + 16]

    def dense_fwd_stream_gram(self, x, w, part, gx_part, w2_part, B, E, K, ldx, ldw, nsplit):
        self.dense_fwd_stream(x, w, part, B, E, K, ldx, ldw, nsplit)
        X = mat(x, B, K, ldx).astype(np.float64)
        W = mat(w, K, E, ldw).astype(np.float64)
        gx = flat(gx_part)[:nsplit * 64 * 64].reshape(nsplit, 64, 64)
        gx[...] = 0
        gx[0, :B, :B] = X @ X.T
        w2 = flat(w2_part)
        w2[:nsplit * (E // 32)] = 0
        w2[0] = (W * W).sum()

    def dense_gram_norm(self, dpre, pre, bias, gx_part, nsplit, w2_part, nw2, l2, partial, nslot, Bk, E, spans=None):
        if spans is not None:
            th, gr, sseg, soff, slen, sl2, spart, nspan = spans
            self.span_sqnorm(th, gr, sseg, soff, slen, sl2, spart, nspan)
        D = mat(dpre, Bk, E, E).astype(np.float64)
        XW = mat(pre, Bk, E, E).astype(np.float64) - flat(bias)[:E].astype(np.float64)
        gx = flat(gx_part)[:nsplit * 64 * 64].reshape(nsplit, 64, 64).astype(np.float64).sum(0)[:Bk, :Bk]
        w2 = float(flat(w2_part)[:nw2].astype(np.float64).sum())
        p = flat(partial)
        p[:2 * nslot] = 0
        p[0] = (gx * (D @ D.T)).sum() + 4 * l2 * (D * XW).sum() + 4 * l2 * l2 * w2
        p[1] = w2

    def enc_tail_fwd_sk_emb(self, part, nsplit, bias, pre, slope, gamma, beta, mov_mean, mov_var, out, xhat, inv_std, rows, C,
                            ldo, training, eps, momentum, r_feat, r_lstm, seed, site_feat, site_lstm, step_dev, emb_table,
                            emb_ids, emb_out, emb_B, emb_T, emb_V, emb_rate, emb_site):
        self.enc_tail_fwd_sk(part, nsplit, bias, pre, slope, gamma, beta, mov_mean, mov_var, out, xhat, inv_std, rows, C, ldo,
                             training, eps, momentum, r_feat, r_lstm, seed, site_feat, site_lstm, step_dev)
        if emb_rate > 0:
            self.embedding_fwd_drop(emb_table, emb_ids, None, emb_out, emb_B, emb_T, C, ldo, emb_V, emb_rate, seed, emb_site, 0,
                                    step_dev)
        else:
            self.embedding_fwd(emb_table, emb_ids, emb_out, emb_B, emb_T, C, ldo, emb_V)

    def enc_tail_fwd_sk(self, part, nsplit, bias, pre, slope, gamma, beta, mov_mean, mov_var, out, xhat, inv_std, rows, C,
                        ldo, training, eps, momentum, r_feat, r_lstm, seed, site_feat, site_lstm, step_dev=None):
        z = flat(part)[:nsplit * rows * C].reshape(nsplit, rows, C).astype(np.float64).sum(0) + flat(bias)[:C]
        mat(pre, rows, C, C)[...] = z
        y = torch.from_numpy(np.where(z > 0, z, z * slope).astype(np.float32))
        self.enc_tail_fwd(y, gamma, beta, mov_mean, mov_var, out, xhat, inv_std, rows, C, ldo, training, eps, momentum,
                          r_feat, r_lstm, seed, site_feat, site_lstm, step_dev)

    def enc_tail_bwd_drop(self, dout, xhat, gamma, inv_std, pre, dpre, dgamma, dbeta, dbias, rows, C, ldo, r_feat, r_lstm,
                          slope, seed, site_feat, site_lstm, step_dev, drop_x, drop_rows, drop_cols, drop_ld, drop_tmajor_B,
                          drop_lwidth, drop_lcol0, drop_rate, drop_site):
        self.enc_tail_bwd(dout, xhat, gamma, inv_std, pre, dpre, dgamma, dbeta, dbias, rows, C, ldo, r_feat, r_lstm, slope,
                          seed, site_feat, site_lstm, step_dev)
        if drop_rate > 0:
            self.dropout(drop_x, drop_x, drop_rows, drop_cols, drop_ld, drop_tmajor_B, drop_lwidth, drop_lcol0, drop_rate, seed,
                         drop_site, 0, step_dev)

    def enc_tail_bwd(self, dout, xhat, gamma, inv_std, pre, dpre, dgamma, dbeta, dbias, rows, C, ldo, r_feat, r_lstm,
                     slope, seed, site_feat, site_lstm, step_dev=None):
        dy = torch.zeros(rows, C, dtype=torch.float32)
        dy.copy_(torch.as_strided(dout, (rows, C), (ldo, 1)))
        if r_lstm > 0:
            self.dropout(dy, dy, rows, C, C, 0, C, 0, r_lstm, seed, site_lstm, 0, step_dev)
        dx = torch.zeros(rows, C, dtype=torch.float32)
        self.batchnorm_bwd(dy, xhat, gamma, inv_std, dx, dgamma, dbeta, rows, C, C, True, None)
        if r_feat > 0:
            self.dropout(dx, dx, rows, C, C, 0, C, 0, r_feat, seed, site_feat, 0, step_dev)
        self.act_bwd(pre, dx, dpre, rows * C, 1, slope)
        self.colsum(dpre, dbias, rows, C, C, None)

    def layernorm_fwd(self, x, gamma, beta, y, xhat, inv_std, rows, C, ldy, eps):
        yo, (xh, inv) = O.layernorm_fwd(mat(x, rows, C, C).astype(np.float64), flat(gamma)[:C].astype(np.float64),
                                        flat(beta)[:C].astype(np.float64), eps)
        mat(y, rows, C, ldy)[...] = yo
        mat(xhat, rows, C, C)[...] = xh
        flat(inv_std)[:rows] = inv[:, 0]

    def layernorm_bwd(self, dy, xhat, gamma, inv_std, dx, dgamma, dbeta, rows, C, lddy, work):
        cache = (mat(xhat, rows, C, C).astype(np.float64), flat(inv_std)[:rows].astype(np.float64)[:, None])
        dxo, dg, db = O.layernorm_bwd(mat(dy, rows, C, lddy).astype(np.float64), flat(gamma)[:C].astype(np.float64), cache)
        if dx is not None:
            mat(dx, rows, C, C)[...] = dxo
        if dgamma is not None:
            flat(dgamma)[:C] = dg
            flat(dbeta)[:C] = db

    def ln_lstm_cell_fwd(self, zk, zr, bias, c_prev, gamma_s, beta_s, gates, chat, istd, c, h, B, U, eps):
        """include/tnt_hip.h: tnt_ln_lstm_cell_fwd_f32 (gate-interleaved [B][U][4] tensors)"""
        z = (flat(zk)[:B * U * 4].astype(np.float64) + flat(zr)[:B * U * 4]).reshape(B, U, 4) + flat(bias)[:U * 4].astype(np.float64).reshape(1, U, 4)
        gi, gf, gg, go = O.sigmoid(z[..., 0]), O.sigmoid(z[..., 1]), np.tanh(z[..., 2]), O.sigmoid(z[..., 3])
        craw = gf * flat(c_prev)[:B * U].astype(np.float64).reshape(B, U) + gi * gg
        cn, (xh, inv) = O.layernorm_fwd(craw, flat(gamma_s)[:U].astype(np.float64), flat(beta_s)[:U].astype(np.float64), eps)
        flat(gates)[:B * U * 4] = np.stack([gi, gf, gg, go], -1).reshape(-1)
        flat(chat)[:B * U] = xh.reshape(-1); flat(istd)[:B] = inv.reshape(-1)
        flat(c)[:B * U] = cn.reshape(-1); flat(h)[:B * U] = (go * np.tanh(cn)).reshape(-1)

    def ln_lstm_cell_bwd(self, dh_a, dh_b, dh_c, dcn_in, gates, c_prev, c, chat, istd, gamma_s, dz, dc_prev, dcnt, B, U):
        f64 = lambda t, *sh: flat(t)[:int(np.prod(sh))].astype(np.float64).reshape(*sh)
        dh = sum(f64(t, B, U) for t in (dh_a, dh_b, dh_c) if t is not None)
        g4 = f64(gates, B, U, 4)
        gi, gf, gg, go = g4[..., 0], g4[..., 1], g4[..., 2], g4[..., 3]
        tc = np.tanh(f64(c, B, U))
        dcn = (f64(dcn_in, B, U) if dcn_in is not None else 0) + dh * go * (1 - tc * tc)
        dcr, _, _ = O.layernorm_bwd(dcn, f64(gamma_s, U), (f64(chat, B, U), f64(istd, B)[:, None]))
        cp = f64(c_prev, B, U)
        out = np.stack([dcr * gg * gi * (1 - gi), dcr * cp * gf * (1 - gf), dcr * gi * (1 - gg * gg), dh * tc * go * (1 - go)], -1)
        flat(dcnt)[:B * U] = dcn.reshape(-1)
        flat(dz)[:B * U * 4] = out.reshape(-1)
        flat(dc_prev)[:B * U] = (dcr * gf).reshape(-1)

    def attention_front_bwd_parts(self, rows, D, A):
        return 1

    def attention_front_bwd(self, Ppre, dP, F, W1, dF, dW1, db1, part, rows, D, A, slope=0.2, drop=None):
        pre = mat(Ppre, rows, A, A).astype(np.float64)
        g = mat(dP, rows, A, A).astype(np.float64) * np.where(pre > 0, 1.0, slope)
        Fm = mat(F, rows, D, D).astype(np.float64)
        W = flat(W1)[:D * A].reshape(D, A).astype(np.float64)
        flat(db1)[:A] = g.sum(0)
        flat(dW1)[:D * A] = (Fm.T @ g).reshape(-1)
        mat(dF, rows, D, D)[...] += g @ W.T
        if drop is not None and drop[0] > 0:
            rate, seed, site, step_dev = drop
            self.dropout(dF, dF, rows, D, D, 0, D, 0, rate, seed, site, 0, step_dev)

    def bias_act_drop_bwd(self, dy, pre, dx, dbias, rows, cols, ld, act, slope, tmajor_B, lwidth, lcol0, rate, seed, site,
                          step_dev=None, extra=None):
        tmp = torch.zeros(rows, cols, dtype=torch.float32)
        tmp.copy_(torch.as_strided(dy, (rows, cols), (ld, 1)))
        if rate > 0:
            self.dropout(tmp, tmp, rows, cols, cols, tmajor_B, lwidth, lcol0, rate, seed, site, 0, step_dev)
        p = torch.zeros(rows, cols, dtype=torch.float32)
        p.copy_(torch.as_strided(pre, (rows, cols), (ld, 1)))
        self.act_bwd(p, tmp, tmp, rows * cols, act, slope)
        torch.as_strided(dx, (rows, cols), (ld, 1)).copy_(tmp)
        self.colsum(tmp, dbias, rows, cols, cols, None)
        if extra is not None:
            self.colsum(*extra, None)

    def colsum_multi(self, jobs):
        for x, out, rows, C, ld in jobs:
            self.colsum(x, out, rows, C, ld, None)

    def colsum2(self, x0, out0, rows0, C0, ld0, x1, out1, rows1, C1, ld1):
        self.colsum(x0, out0, rows0, C0, ld0, None)
        self.colsum(x1, out1, rows1, C1, ld1, None)

    def colsum(self, x, out, rows, C, ld, work):
        flat(out)[:C] = mat(x, rows, C, ld).astype(np.float64).sum(0)

    def dense_dw_skinny(self, x, dpre, dw, N, E, Bk, ldx):
        xs = mat(x, Bk, N, ldx).astype(np.float64)
        mat(dw, N, E, E)[...] = xs.T @ mat(dpre, Bk, E, E).astype(np.float64)

    def sum2(self, x0, out0, x1, out1, n, scale):
        self.sum(x0, out0, n, scale)
        self.sum(x1, out1, n, scale)

    def stage_batch(self, x, x_dst, cap, cap_dst, tgt, tgt_tmajor, a0, h0, c0, c0_dst, B, T, N, ldx, U, xT_dst=None, ldt=0,
                    masks=None):
        if masks is not None:
            self.dropout_mask4(masks[0], masks[1], masks[2], masks[3], masks[4], masks[5], 0, masks[6])
        xs = x.reshape(-1)[:B * N].float().numpy().reshape(B, N)          # float32 or float16 ("on-wire") betas
        mat(x_dst, B, N, ldx)[...] = xs
        if xT_dst is not None:
            mat(xT_dst, N, B, ldt)[...] = xs.T
        flat(cap_dst)[:B * T] = flat(cap)[:B * T]
        if tgt is not None:
            flat(tgt_tmajor)[:B * T] = flat(tgt)[:B * T].reshape(B, T).T.reshape(-1)
        flat(h0)[:B * U] = flat(a0)[:B * U]
        flat(c0_dst)[:B * U] = flat(c0)[:B * U]

    def sum(self, x, out, n, scale):
        flat(out)[0] = flat(x)[:n].astype(np.float64).sum() * scale

    # ---------------------------------------------------------------- embedding
    def embedding_fwd(self, table, ids, out, B, T, E, ldo, V):
        idv = np.clip(flat(ids)[:B * T].reshape(B, T), 0, V - 1)
        tab = mat(table, V, E, E)
        mat(out, T * B, E, ldo)[...] = tab[idv.T.reshape(-1)]

    def embedding_fwd_drop(self, table, ids, out, out_drop, B, T, E, ldo, V, rate, seed, site, step, step_dev=None,
                           mask2=None):
        tmp = out if out is not None else torch.zeros(B * T, ldo, dtype=torch.float32)
        self.embedding_fwd(table, ids, tmp, B, T, E, ldo, V)
        self.dropout(tmp, out_drop, T * B, E, ldo, B, E, 0, rate, seed, site, step, step_dev)
        if mask2 is not None and mask2[0] > 0:
            rate2, site2, lw2, lc2 = mask2
            self.dropout(out_drop, out_drop, T * B, E, ldo, 0, lw2, lc2, rate2, seed, site2, step, step_dev, rows_per_site=B)

    def embedding_bwd(self, drows, ids, dtable, sq_norm, rowsq_work, B, T, E, ldd, V):
        idv = flat(ids)[:B * T].reshape(B, T)
        rows = mat(drows, T * B, E, ldd).astype(np.float64)
        g = np.zeros((V, E))
        np.add.at(g, idv.T.reshape(-1), rows)
        mat(dtable, V, E, E)[...] = g
        if sq_norm is not None:
            flat(sq_norm)[0] = (rows * rows).sum()

    # ---------------------------------------------------------------- LSTM
    def lstm_step_fwd(self, xz, h_prev, c_prev, Ur, ctx, Wc, D, mask_ids, mask_T, mask_t, out_prev, h, c, out,
                      gates, B, U, xz_bias=None):
        z = mat(xz, B, 4 * U, 4 * U).astype(np.float64).reshape(B, U, 4)
        if xz_bias is not None:
            z = z + flat(xz_bias)[:4 * U].astype(np.float64).reshape(1, U, 4)
        hp, cp = mat(h_prev, B, U, U).astype(np.float64), mat(c_prev, B, U, U).astype(np.float64)
        z = z + (hp @ mat(Ur, U, 4 * U, 4 * U).astype(np.float64)).reshape(B, U, 4)
        if ctx is not None:
            z = z + (mat(ctx, B, D, D).astype(np.float64) @ mat(Wc, D, 4 * U, 4 * U).astype(np.float64)).reshape(B, U, 4)
        i, f, g, o = O.sigmoid(z[..., 0]), O.sigmoid(z[..., 1]), np.tanh(z[..., 2]), O.sigmoid(z[..., 3])
        c2 = f * cp + i * g
        h2 = o * np.tanh(c2)
        m = np.ones((B, 1), bool)
        if mask_ids is not None:
            m = (flat(mask_ids)[:B * mask_T].reshape(B, mask_T)[:, mask_t] != 0)[:, None]
        op = mat(out_prev, B, U, U) if out_prev is not None else 0.0
        mat(h, B, U, U)[...] = np.where(m, h2, hp)
        mat(c, B, U, U)[...] = np.where(m, c2, cp)
        if out is not None:
            mat(out, B, U, U)[...] = np.where(m, h2, op)
        mat(gates, B, 4 * U, 4 * U)[...] = np.stack([i, f, g, o], -1).reshape(B, 4 * U)

    def lstm_step_bwd(self, dz_next, Ur, da_pass_in, dh_ext, dc_in, dout_in, dout_t, mask_ids, mask_T, mask_t, gates,
                      c, c_prev, dz, da_pass_out, dc_out, dout_out, B, U, Wc=None, D=0, dctx_part=None):
        g64 = lambda t: mat(t, B, U, U).astype(np.float64) if t is not None else np.zeros((B, U))
        da = g64(da_pass_in) + g64(dh_ext)
        if dz_next is not None:
            da = da + mat(dz_next, B, 4 * U, 4 * U).astype(np.float64) @ mat(Ur, U, 4 * U, 4 * U).astype(np.float64).T
        dout = g64(dout_in) + g64(dout_t)
        dcin = g64(dc_in)
        m = np.ones((B, 1), bool)
        if mask_ids is not None:
            m = (flat(mask_ids)[:B * mask_T].reshape(B, mask_T)[:, mask_t] != 0)[:, None]
        gt = mat(gates, B, 4 * U, 4 * U).astype(np.float64).reshape(B, U, 4)
        gi, gf, gg, go = gt[..., 0], gt[..., 1], gt[..., 2], gt[..., 3]
        tc = np.tanh(g64(c))
        dh = da + dout
        dgo = dh * tc
        dcc = dcin + dh * go * (1 - tc * tc)
        dzv = np.stack([dcc * gg * gi * (1 - gi), dcc * g64(c_prev) * gf * (1 - gf), dcc * gi * (1 - gg * gg),
                        dgo * go * (1 - go)], -1)
        mat(dz, B, 4 * U, 4 * U)[...] = np.where(m[..., None], dzv, 0).reshape(B, 4 * U)
        if dctx_part is not None:           # per-unit-block partials of dz @ Wc^T
            dzf = mat(dz, B, 4 * U, 4 * U).astype(np.float64).reshape(B, U // 16, 64)
            wc = flat(Wc)[:D * 4 * U].astype(np.float64).reshape(D, U // 16, 64)
            flat(dctx_part)[:(U // 16) * B * D] = np.einsum("bpk,dpk->pbd", dzf, wc).reshape(-1)
        if dc_out is not None:
            mat(dc_out, B, U, U)[...] = np.where(m, dcc * gf, dcin)
        if da_pass_out is not None:
            mat(da_pass_out, B, U, U)[...] = np.where(m, 0, da)
        if dout_out is not None:
            mat(dout_out, B, U, U)[...] = np.where(m, 0, dout)

    # ---------------------------------------------------------------- softmax / CE
    @staticmethod
    def _gru3(t, lead, U):
        """interleaved [lead][U][4] (slots z, r, h, pad) -> keras [lead][3U]"""
        a = flat(t)[:lead * U * 4].reshape(lead, U, 4).astype(np.float64)
        return np.concatenate([a[:, :, 0], a[:, :, 1], a[:, :, 2]], axis=1)

    @staticmethod
    def _gru3_store(t, arr, lead, U):
        out = np.zeros((lead, U, 4), np.float32)
        for g in range(3):
            out[:, :, g] = arr[:, g * U:(g + 1) * U]
        flat(t)[:lead * U * 4] = out.reshape(-1)

    def gru_step_fwd(self, xz, h_prev, Uk, br, h, gates, B, U):
        hp = mat(h_prev, B, U, U).astype(np.float64)
        h2, (z, r, hh, rech, _) = O.gru_step_fwd(self._gru3(xz, B, U), hp, self._gru3(Uk, U, U), self._gru3(br, 1, U)[0])
        mat(h, B, U, U)[...] = h2
        flat(gates)[:B * U * 4] = np.stack([z, r, hh, rech], axis=-1).astype(np.float32).reshape(-1)

    def gru_step_bwd(self, drec_next, Uk, dh_pass_in, dh_ext, gates, h_prev, dxz, drec, dh_pass_out, B, U):
        Ukk = self._gru3(Uk, U, U)
        dh = np.zeros((B, U))
        if dh_pass_in is not None:
            dh += mat(dh_pass_in, B, U, U)
        if dh_ext is not None:
            dh += mat(dh_ext, B, U, U)
        if drec_next is not None:
            dh += self._gru3(drec_next, B, U) @ Ukk.T
        g = flat(gates)[:B * U * 4].reshape(B, U, 4).astype(np.float64)
        cache = (g[:, :, 0], g[:, :, 1], g[:, :, 2], g[:, :, 3], mat(h_prev, B, U, U).astype(np.float64))
        dx, dr, _ = O.gru_step_bwd(dh, cache, Ukk)
        self._gru3_store(dxz, dx, B, U)
        self._gru3_store(drec, dr, B, U)
        if dh_pass_out is not None:
            mat(dh_pass_out, B, U, U)[...] = dh * cache[0]

    def softmax_cce(self, logits, target, probs, loss_row, correct_row, dlogits, rows, V, ld, gscale,
                    from_logits=False, mask_zero=False):
        x = mat(logits, rows, V, ld).astype(np.float64)
        p = O.softmax(x)
        if target is not None:
            y = flat(target)[:rows].astype(np.int64)
            py = np.take_along_axis(p, y[:, None], 1)[:, 0]
            live = (y != 0) if mask_zero else np.ones(rows, bool)
            if loss_row is not None:
                l = O.sparse_cce_from_logits(x, y)[0] if from_logits else -np.log(np.clip(py, 1e-7, 1 - 1e-7))
                flat(loss_row)[:rows] = np.where(live, l, 0.0)
            if correct_row is not None:
                flat(correct_row)[:rows] = (p.argmax(-1) == y)
            if dlogits is not None:
                oh = np.zeros_like(p)
                np.put_along_axis(oh, y[:, None], 1.0, 1)
                active = (live & (from_logits | ((py >= 1e-7) & (py <= 1 - 1e-7))))[:, None]
                mat(dlogits, rows, V, ld)[...] = np.where(active, (p - oh) * gscale, 0)
        if probs is not None and (dlogits is None or probs.data_ptr() != dlogits.data_ptr()):
            mat(probs, rows, V, ld)[...] = p

    def onehot_argmax(self, onehot, ids_tmajor, B, T, V):
        oh = flat(onehot)[:B * T * V].reshape(B, T, V)
        flat(ids_tmajor)[:B * T] = oh.argmax(-1).T.reshape(-1)

    def argmax_rows(self, x, out, rows, V, ld):
        flat(out)[:rows] = mat(x, rows, V, ld).argmax(-1)

    # ---------------------------------------------------------------- optimizer
    def _segs(self, span_seg, span_off, span_len, nspan):
        ss, so, sl = flat(span_seg)[:nspan], flat(span_off)[:nspan], flat(span_len)[:nspan]
        return list(zip(ss.tolist(), so.tolist(), sl.tolist()))

    def seg_sqnorm(self, theta, grad, span_seg, span_off, span_len, seg_first, seg_l2, partial, sq, wsq, l2_out,
                   nspan, nseg):
        th, gr, l2 = flat(theta), flat(grad), flat(seg_l2)
        q, w = np.zeros(nseg), np.zeros(nseg)
        first = flat(seg_first)[:nseg + 1]          # like the kernel: spans -> segment via seg_first (slice-local)
        for k, (s, o, n) in enumerate(self._segs(span_seg, span_off, span_len, nspan)):
            t = th[o:o + n].astype(np.float64)
            g = gr[o:o + n].astype(np.float64) + 2 * float(l2[s]) * t
            loc = int(np.searchsorted(first, k, side="right")) - 1
            q[loc] += (g * g).sum()
            w[loc] += (t * t).sum()
        flat(sq)[:nseg] = q
        flat(wsq)[:nseg] = w
        if l2_out is not None:
            flat(l2_out)[0] = (l2[:nseg].astype(np.float64) * w).sum()

    def sample_rows(self, x, out, rows, V, ld, temperature, from_logits, seed, site, step, step_dev=None):
        st = step + (int(flat(step_dev)[0]) if step_dev is not None else 0)
        ids, _ = O.sample_rows(flat(x)[:rows * ld].reshape(rows, ld)[:, :V], temperature, from_logits, seed, site, st)
        flat(out)[:rows] = ids

    def l2_total(self, wsq, seg_l2, nseg, out):
        flat(out)[0] = (flat(seg_l2)[:nseg].astype(np.float64) * flat(wsq)[:nseg].astype(np.float64)).sum()

    def _clip(self, sq, sq_override, s, clipnorm):
        if clipnorm <= 0:
            return 1.0
        q = float(flat(sq)[s])
        if sq_override is not None and float(flat(sq_override)[s]) >= 0:
            q = float(flat(sq_override)[s])
        return clipnorm / max(np.sqrt(q), clipnorm)

    def adam(self, theta, m, v, grad, span_seg, span_off, span_len, seg_l2, sq, sq_override, nspan, lr_t, lr_t_dev,
             beta1, beta2, eps, clipnorm, guard=None, met=None, ring=None, ring_t=None):
        if ring is not None:              # the metrics ring job (before the guard check, like the kernel)
            rt = int(flat(ring_t)[0])
            nmet = met.numel()
            row = flat(ring)[(rt % ring.shape[0]) * (nmet + 1):][:nmet + 1]
            row[:nmet] = flat(met)[:nmet]
            row[nmet] = float(rt & 0xFFFFFF)
            flat(ring_t)[0] = rt + 1
        if guard is not None and int(guard[0]) != 0:
            return
        if lr_t_dev is not None:
            lr_t = float(flat(lr_t_dev)[0])
        th, mm, vv, gr, l2 = flat(theta), flat(m), flat(v), flat(grad), flat(seg_l2)
        for s, o, n in self._segs(span_seg, span_off, span_len, nspan):
            t = th[o:o + n].astype(np.float64)
            g = (gr[o:o + n].astype(np.float64) + 2 * float(l2[s]) * t) * self._clip(sq, sq_override, s, clipnorm)
            m1 = mm[o:o + n] + (g - mm[o:o + n]) * (1 - beta1)
            v1 = vv[o:o + n] + (g * g - vv[o:o + n]) * (1 - beta2)
            th[o:o + n] = t - lr_t * m1 / (np.sqrt(v1) + eps)
            mm[o:o + n] = m1
            vv[o:o + n] = v1

    def dense_dw_sqnorm(self, x, dpre, theta, l2, partial, nslot, N, E, Bk, ldx):
        g = mat(x, Bk, N, ldx).astype(np.float64).T @ mat(dpre, Bk, E, E).astype(np.float64)
        t = flat(theta)[:N * E].reshape(N, E).astype(np.float64)
        p = flat(partial)
        p[:2 * nslot] = 0
        p[0], p[1] = ((g + 2 * l2 * t) ** 2).sum(), (t * t).sum()

    def dense_dw_adam(self, x, dpre, theta, m, v, l2, sq, sq_override, lr_t_dev, beta1, beta2, eps, clipnorm, N, E, Bk, ldx,
                      guard=None):
        if guard is not None and int(guard[0]) != 0:
            return
        lr_t = float(flat(lr_t_dev)[0])
        g = (mat(x, Bk, N, ldx).astype(np.float64).T @ mat(dpre, Bk, E, E).astype(np.float64)).reshape(-1)
        th, mm, vv = flat(theta), flat(m), flat(v)
        n = N * E
        t = th[:n].astype(np.float64)
        g = (g + 2 * l2 * t) * self._clip(sq, sq_override, 0, clipnorm)
        m1 = mm[:n] + (g - mm[:n]) * (1 - beta1)
        v1 = vv[:n] + (g * g - vv[:n]) * (1 - beta2)
        th[:n] = t - lr_t * m1 / (np.sqrt(v1) + eps)
        mm[:n] = m1
        vv[:n] = v1

    def sgd(self, theta, mom, grad, span_seg, span_off, span_len, seg_l2, sq, sq_override, nspan, lr, lr_dev, momentum,
            clipnorm, guard=None):
        if guard is not None and int(guard[0]) != 0:
            return
        if lr_dev is not None:
            lr = float(flat(lr_dev)[0])
        th, mo, gr, l2 = flat(theta), flat(mom), flat(grad), flat(seg_l2)
        for s, o, n in self._segs(span_seg, span_off, span_len, nspan):
            t = th[o:o + n].astype(np.float64)
            g = (gr[o:o + n].astype(np.float64) + 2 * float(l2[s]) * t) * self._clip(sq, sq_override, s, clipnorm)
            mv = momentum * mo[o:o + n] - lr * g
            mo[o:o + n] = mv
            th[o:o + n] = t + mv

    def sqdiff_mean(self, x, out, n, c):
        flat(out)[0] = ((c - flat(x)[:n].astype(np.float64)) ** 2).mean()

    def sam(self, theta, grad, ew, span_seg, span_off, span_len, seg_l2, sq, nseg, nspan, rho, mode, sq_override=None):
        th, gr, e, l2 = flat(theta), flat(grad), flat(ew), flat(seg_l2)
        q = flat(sq)[:nseg].astype(np.float64).copy()
        if sq_override is not None:
            o = flat(sq_override)[:nseg].astype(np.float64)
            q = np.where(o >= 0, o, q)
        scale = rho / (np.sqrt(q.sum()) + 1e-12)
        for s, o, n in self._segs(span_seg, span_off, span_len, nspan):
            if mode == 1:
                th[o:o + n] -= e[o:o + n]
                gr[o:o + n] += 2 * float(l2[s]) * e[o:o + n]
            else:
                t = th[o:o + n].astype(np.float64)
                e[o:o + n] = (gr[o:o + n].astype(np.float64) + 2 * float(l2[s]) * t) * scale
                th[o:o + n] = t + e[o:o + n]

    def beam_topk(self, probs, score_in, fin_in, B, V, ld, k, end_id, score_out, parent, token, fin_out):
        """include/tnt_hip.h: tnt_beam_topk_f32, from its definition (float32 scores like the kernel)."""
        P = mat(probs, B * k, V, ld)
        sc, fn = flat(score_in)[:B * k].reshape(B, k), flat(fin_in)[:B * k].reshape(B, k)
        so, pa, to, fo = flat(score_out), flat(parent), flat(token), flat(fin_out)
        for b in range(B):
            cand = np.empty((k, V), np.float32)
            for j in range(k):
                if fn[b, j]:
                    cand[j] = -np.inf; cand[j, 0] = sc[b, j]
                else:
                    cand[j] = np.float32(sc[b, j]) + np.log(np.maximum(P[b * k + j], np.float32(1e-30))).astype(np.float32)
            order = np.argsort(-cand.reshape(-1), kind="stable")[:k]
            for r, cnd in enumerate(order):
                j, v = divmod(int(cnd), V)
                so[b * k + r] = cand[j, v]; pa[b * k + r] = b * k + j; to[b * k + r] = v
                fo[b * k + r] = 1 if (fn[b, j] or v == end_id) else 0

    def block_dense_dx(self, dpre, W, dx, B, R, Din, Dout):
        d = flat(dpre)[:B * R * Dout].reshape(B, R, Dout).astype(np.float64)
        w = flat(W)[:R * Din * Dout].reshape(R, Din, Dout).astype(np.float64)
        flat(dx)[:B * R * Din] = np.einsum("brn,rkn->brk", d, w).reshape(-1)

    def embedding_bwd_parts(self, B, T, E):
        return B * T * ((E + 255) // 256)

    def embedding_bwd_sparse(self, drows, ids, prev_ids, dtable, sq_part, B, T, E, ldd, V, drop_rate=0.0, drop_seed=0,
                             drop_site=0, drop_step_dev=None, zero_id=-1):
        """include/tnt_hip.h: tnt_embedding_bwd_sparse_f32 from its contract: only rows of ids and prev_ids are touched"""
        idv = np.clip(flat(ids)[:B * T].reshape(B, T), 0, V - 1)
        if zero_id >= 0:          # the caller's guarantee, checked here: the rows the kernel will not read ARE zero
            r = mat(drows, T * B, E, ldd).reshape(T, B, E)
            assert not r[(idv == zero_id).T].any(), "embedding_bwd_sparse: rows of zero_id carry a gradient"
        if drop_rate > 0:         # the mask the forward applied, on a copy (the row buffer itself stays undropped)
            tmp = torch.zeros(T * B, E, dtype=torch.float32)
            tmp.copy_(torch.as_strided(drows, (T * B, E), (ldd, 1)))
            self.dropout(tmp, tmp, T * B, E, E, B, E, 0, drop_rate, drop_seed, drop_site, 0, drop_step_dev)
            drows, ldd = tmp, E
        rows = mat(drows, T * B, E, ldd).reshape(T, B, E).astype(np.float64)
        tab = flat(dtable)[:V * E].reshape(V, E)
        for p in set(int(x) for x in flat(prev_ids)[:B * T] if 0 <= x < V):
            tab[p] = 0
        acc = {}
        for b in range(B):
            for t in range(T):
                acc[int(idv[b, t])] = acc.get(int(idv[b, t]), 0) + rows[t, b]
        for i, v in acc.items():
            tab[i] = v
        ny = (E + 255) // 256
        sp = flat(sq_part)
        sp[:B * T * ny] = 0
        sp[0] = (rows * rows).sum()

    def span_sqnorm(self, theta, grad, span_seg, span_off, span_len, seg_l2, partial, nspan):
        th, gr, l2, pa = flat(theta), flat(grad), flat(seg_l2), flat(partial)
        for k, (s, o, n) in enumerate(self._segs(span_seg, span_off, span_len, nspan)):
            t = th[o:o + n].astype(np.float64)
            g = gr[o:o + n].astype(np.float64) + 2 * float(l2[s]) * t
            pa[2 * k], pa[2 * k + 1] = (g * g).sum(), (t * t).sum()

    def step_finalize(self, partial, seg_first, seg_l2, sq, wsq, l2_out, nseg, x0=None, out0=None, x1=None, out1=None, n=0,
                      scale=1.0, extra_part=None, extra=None, n_extra=0, ids_src=None, ids_dst=None, n_ids=0, adam_t=None,
                      drop_step=None, lr=None, lr_t=None, beta1=0.0, beta2=0.0, guard=None, x2=None, out2=None, n2=0,
                      scale2=1.0):
        if n2 > 0:
            flat(out2)[0] = flat(x2)[:n2].astype(np.float64).sum() * scale2
        if nseg > 0:
            pa, first = flat(partial).astype(np.float64), flat(seg_first)
            for s in range(nseg):
                k0, k1 = int(first[s]), int(first[s + 1])
                flat(sq)[s] = pa[2 * k0:2 * k1:2].sum()
                flat(wsq)[s] = pa[2 * k0 + 1:2 * k1:2].sum()
            if l2_out is not None:
                flat(l2_out)[0] = (flat(seg_l2)[:nseg].astype(np.float64) * flat(wsq)[:nseg].astype(np.float64)).sum()
        if n > 0:
            flat(out0)[0] = flat(x0)[:n].astype(np.float64).sum() * scale
            if x1 is not None:
                flat(out1)[0] = flat(x1)[:n].astype(np.float64).sum() * scale
        if n_extra > 0:
            flat(extra)[0] = flat(extra_part)[:n_extra].astype(np.float64).sum()
        if n_ids > 0 and ids_src is not None and ids_dst is not None:
            flat(ids_dst)[:n_ids] = flat(ids_src)[:n_ids]
        self.step_tick(adam_t, drop_step, lr, lr_t, beta1, beta2, guard=guard)

    def colsq(self, x, out, rows, cols, ld):
        m = mat(x, rows, cols, ld).astype(np.float64)
        flat(out)[:cols] = (m * m).sum(0)

    def agc(self, theta, grad, tab, gsq_cols=None, sq_out=None, clip_factor=0.01, eps=1e-3):
        """include/tnt_hip.h: tnt_agc_f32, straight from the definition (per variable and unit column)."""
        th, gr = flat(theta), flat(grad)
        items = tab.item.numpy().reshape(-1, 6)
        off, ld, lam = tab.var_off.numpy(), tab.var_ld.numpy(), tab.var_lam.numpy()
        sq_total = 0.0
        for v in np.unique(items[:, 0]):
            it = items[items[:, 0] == v]
            rows, cols = int(it[:, 4].max()), int(ld[v])
            o, l2 = int(off[v]), 2 * float(lam[v])
            W = th[o:o + rows * cols].reshape(rows, cols).astype(np.float64)
            G = gr[o:o + rows * cols].reshape(rows, cols).astype(np.float64) + l2 * W
            pn = np.sqrt((W * W).sum(0))
            gsq = (G * G).sum(0)
            emb = gsq_cols is not None and tab.emb is not None and v == tab.emb[0]
            if emb:
                gsq = flat(gsq_cols)[:cols].astype(np.float64)
            gn = np.sqrt(gsq)
            mx = np.maximum(pn, eps) * clip_factor
            sc = np.where(gn < mx, 1.0, mx / np.maximum(gn, 1e-6))
            gr[o:o + rows * cols] = ((G * sc) - l2 * W).reshape(-1)
            if emb:
                sq_total = float((sc * sc * gsq).sum())
        if gsq_cols is not None and sq_out is not None:
            flat(sq_out)[0] = sq_total

    def step_tick(self, adam_t, drop_step, lr, lr_t, beta1, beta2, guard=None):
        if guard is not None and int(guard[0]) != 0:
            return
        if drop_step is not None:
            drop_step += 1
        if adam_t is not None:
            adam_t += 1
            t = int(adam_t[0])
            if lr_t is not None:
                flat(lr_t)[0] = float(flat(lr)[0]) * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)

    # ---------------------------------------------------------------- encoder / attention
    def _groups(self, idx, goff, R):
        go = flat(goff)[:R + 1]
        ix = flat(idx)
        return [ix[go[r]:go[r + 1]].astype(np.int64) for r in range(R)], go

    def locally_dense_fwd(self, x, ldx, idx, goff, W, bias, pre, y, B, R, D, slope=0.2):
        groups, go = self._groups(idx, goff, R)
        xs = mat(x, B, int(max(g.max() for g in groups if len(g)) + 1), ldx).astype(np.float64)
        Wm = mat(W, int(go[R]), D, D).astype(np.float64)
        bm = mat(bias, R, D, D).astype(np.float64)
        yo, po = O.locally_dense_fwd(xs, groups, [Wm[go[r]:go[r + 1]] for r in range(R)], list(bm), slope)
        flat(pre)[:B * R * D] = po.reshape(-1)
        flat(y)[:B * R * D] = yo.reshape(-1)

    def locally_dense_bwd(self, x, ldx, idx, goff, dpre, dW, db, B, R, D):
        groups, go = self._groups(idx, goff, R)
        xs = mat(x, B, int(max(g.max() for g in groups if len(g)) + 1), ldx).astype(np.float64)
        dp = flat(dpre)[:B * R * D].reshape(B, R, D).astype(np.float64)
        dWm = mat(dW, int(go[R]), D, D)
        for r, g in enumerate(groups):
            dWm[go[r]:go[r + 1]] = xs[:, g].T @ dp[:, r]
        mat(db, R, D, D)[...] = dp.sum(0)

    def attention_step_fwd(self, h, F, P, W2, b2, v, bv, qpre, alpha, ctx, ctx_d, s_out, B, R, D, A, U, slope,
                           rate_attn, rate_in, in_lwidth, seed, site_attn, site_in, step, step_dev=None, keep4=None):
        if step_dev is not None:
            step = (step + int(step_dev[0])) & 0xFFFFFFFF
        f64 = lambda t, *s: flat(t)[:int(np.prod(s))].reshape(*s).astype(np.float64)
        keep = _keep(np.arange(B * R * A).reshape(B, R, A), rate_attn, seed, site_attn, step) if rate_attn > 0 else None
        self._check_keep4(keep4, keep, B * R * A)
        (cx, al, sd), cache = O.attention_step_fwd(f64(h, B, U), f64(F, B, R, D), f64(P, B, R, A), f64(W2, U, A),
                                                   f64(b2, A), f64(v, A)[:, None], f64(bv, 1), keep, rate_attn, slope)
        flat(qpre)[:B * A] = cache[1].reshape(-1)
        flat(alpha)[:B * R] = al.reshape(-1)
        flat(ctx)[:B * D] = cx.reshape(-1)
        if s_out is not None:
            flat(s_out)[:B * R * A] = sd.reshape(-1)
        if ctx_d is not None:
            kin = _keep(np.arange(B)[:, None] * in_lwidth + np.arange(D)[None, :], rate_in, seed, site_in, step) if rate_in > 0 else None
            flat(ctx_d)[:B * D] = O.dropout_fwd(cx, kin, rate_in).reshape(-1)

    def attention_step_bwd(self, dctx_d, F, P, W2, v, qpre, alpha, dP, dF, dvb, dqpre, dh, B, R, D, A, U, slope,
                           rate_attn, rate_in, in_lwidth, seed, site_attn, site_in, step, step_dev=None, dz=None,
                           Wc=None, dctx_part=None, nparts=0, keep4=None, alpha_mse=0.0, fresh=False):
        if step_dev is not None:
            step = (step + int(step_dev[0])) & 0xFFFFFFFF
        f64 = lambda t, *s: flat(t)[:int(np.prod(s))].reshape(*s).astype(np.float64)
        keep = _keep(np.arange(B * R * A).reshape(B, R, A), rate_attn, seed, site_attn, step) if rate_attn > 0 else None
        self._check_keep4(keep4, keep, B * R * A)
        kin = _keep(np.arange(B)[:, None] * in_lwidth + np.arange(D)[None, :], rate_in, seed, site_in, step) if rate_in > 0 else None
        if dctx_part is not None:
            raw = f64(dctx_part, nparts, B, D).sum(0)
        else:
            raw = f64(dz, B, 4 * U) @ f64(Wc, D, 4 * U).T if dz is not None else f64(dctx_d, B, D)
        dctx = O.dropout_bwd(raw, kin, rate_in)
        Fm, Pm, W2m, vm, qp, al = f64(F, B, R, D), f64(P, B, R, A), f64(W2, U, A), f64(v, A)[:, None], f64(qpre, B, A), f64(alpha, B, R)
        q = O.act_fwd(qp, O.ACT_LEAKY, slope)
        s = np.tanh(Pm + q[:, None, :])
        sd = O.dropout_fwd(s, keep, rate_attn)
        cache = (None, qp, s, sd, al, keep, rate_attn)
        dalpha = (dctx[:, None, :] * Fm).sum(2) + alpha_mse * (al - 1.0)
        dFv = al[:, :, None] * dctx[:, None, :]
        de = al * (dalpha - (al * dalpha).sum(1, keepdims=True))
        dv = (sd * de[:, :, None]).sum(1)                     # per-sample (B,A)
        dsum = O.dropout_bwd(de[:, :, None] * vm[:, 0], keep, rate_attn) * (1 - s * s)
        dq = O.act_bwd(qp, dsum.sum(1), O.ACT_LEAKY, slope)
        if fresh:
            flat(dP)[:B * R * A] = 0
            flat(dF)[:B * R * D] = 0
            flat(dvb)[:B * (A + 1)] = 0
        flat(dP)[:B * R * A] += dsum.reshape(-1).astype(np.float32)
        flat(dF)[:B * R * D] += dFv.reshape(-1).astype(np.float32)
        dvbm = flat(dvb)[:B * (A + 1)].reshape(B, A + 1)
        dvbm[:, :A] += dv
        dvbm[:, A] += de.sum(1)
        flat(dqpre)[:B * A] = dq.reshape(-1)
        flat(dh)[:B * U] = (dq @ W2m.T).reshape(-1)

    def attention_metric(self, alpha, out, work, T, B, R, tstride=0):
        ts = tstride if tstride > 0 else B * R
        al = np.lib.stride_tricks.as_strided(flat(alpha), (T, B, R), (ts * 4, R * 4, 4)).astype(np.float64)
        v = ((1 - al.sum(1)) ** 2)
        if out is None:                       # partials only: one per timestep, the rest of the slots zero
            n = self.attention_metric_parts(T, R)
            flat(work)[:n] = 0
            flat(work)[:T] = v.sum(1)
            return
        flat(out)[0] = v.mean()

    def attention_metric_parts(self, T, R):
        return T * ((R + 63) // 64)

"""Per-kernel parity: every C-ABI entry point against the numpy oracle (float64 truth),
called through the ctypes binding on a real MI355X.  Tolerance: 1e-4 relative (the
north-star bound on logits) unless noted; integer/index outputs bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle.philox import keep_mask
from helpers import tiny_groups

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


def close(got, want, rtol=RTOL, atol=None):
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    scale = np.abs(want).max() + 1e-30
    atol = rtol * scale if atol is None else atol
    err = np.abs(got - want).max()
    assert err <= atol, f"max abs err {err:.3e} > {atol:.3e} (scale {scale:.3e})"


@pytest.fixture(scope="module")
def be():
    import masters_thesis_amd.ops as ops
    return ops.backend()


def il(w, U):
    """keras [.., 4U] gate blocks -> interleaved [.., U, 4]."""
    lead = w.shape[:-1]
    return np.ascontiguousarray(np.moveaxis(w.reshape(*lead, 4, U), -2, -1))


def unil(w):
    """interleaved [.., U, 4] -> keras [.., 4U]."""
    lead = w.shape[:-2]
    return np.ascontiguousarray(np.moveaxis(w, -1, -2)).reshape(*lead, -1)


# ------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K,tA,tB,pad", [
    (64, 512, 2000, 0, 0, 0), (960, 5001, 512, 0, 0, 3), (960, 512, 5001, 0, 1, 3), (512, 5001, 960, 1, 0, 3),
    (33, 17, 29, 0, 0, 0), (33, 17, 29, 0, 1, 0), (33, 17, 29, 1, 0, 0), (200, 130, 64, 1, 0, 2),
    (1024, 2048, 512, 0, 0, 0), (5, 7, 3, 0, 0, 0), (5, 7, 3, 0, 0, 1),
])
def test_gemm(be, M, N, K, tA, tB, pad):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((M, K))
    Bm = rng.standard_normal((K, N))
    bias = rng.standard_normal(N)
    As = A.T if tA else A
    Bs = Bm.T if tB else Bm
    lda, ldb, ldc = As.shape[1] + pad, Bs.shape[1] + pad, N + pad
    Ad = torch.zeros(As.shape[0], lda, device="cuda"); Ad[:, :As.shape[1]] = dev(As)
    Bd = torch.zeros(Bs.shape[0], ldb, device="cuda"); Bd[:, :Bs.shape[1]] = dev(Bs)
    Cd = torch.full((M, ldc), 7.0, device="cuda")
    Pd = torch.zeros(M, ldc, device="cuda")
    be.gemm(Ad, Bd, Cd, M, N, K, lda, ldb, ldc, bool(tA), bool(tB), bias=dev(bias), pre=Pd, act=1, slope=0.2)
    pre = A @ Bm + bias
    close(Pd[:, :N], pre)
    close(Cd[:, :N], np.where(pre > 0, pre, 0.2 * pre))
    if pad:
        assert (Cd[:, N:] == 7.0).all()      # padding untouched
    # accumulate
    C2 = torch.ones(M, ldc, device="cuda")
    be.gemm(Ad, Bd, C2, M, N, K, lda, ldb, ldc, bool(tA), bool(tB), accumulate=True)
    close(C2[:, :N], A @ Bm + 1.0)


@pytest.mark.parametrize("M,N,K,pad,forced", [(960, 5001, 512, 3, False), (960, 5001, 256, 3, False),
                                               (100, 70, 44, 2, True), (330, 257, 96, 3, True), (161, 129, 36, 3, True),
                                               (16, 16, 8, 0, True), (480, 300, 1000, 0, True)])
def test_gemm_one_round(be, M, N, K, pad, forced):
    """The one-round 160x128 kernel (16x16x4 MFMAs, one workgroup per CU): picked by tnt_gemm_f32 itself on the
    vocabulary-head shapes, forced through tnt_gemm_f32_tile on ragged ones (row / column / k tails, every epilogue
    mode), against float64 numpy and against the tiled kernel."""
    rng = np.random.default_rng(M * 11 + N * 5 + K)
    A, Bm, bias = rng.standard_normal((M, K)), rng.standard_normal((K, N)), rng.standard_normal(N)
    lda, ldb, ldc = K, N + pad, N + pad
    assert ldb % 4 == 0 and lda % 4 == 0
    Ad = dev(A)
    Bd = torch.zeros(K, ldb, device="cuda"); Bd[:, :N] = dev(Bm)
    run = (lambda *a, **k: be.gemm_tile(*a, 160, 128, **k)) if forced else be.gemm
    Cd = torch.full((M, ldc), 7.0, device="cuda"); Pd = torch.zeros(M, ldc, device="cuda")
    run(Ad, Bd, Cd, M, N, K, lda, ldb, ldc, bias=dev(bias), pre=Pd, act=1, slope=0.2)
    pre = A @ Bm + bias
    close(Pd[:, :N], pre)
    close(Cd[:, :N], np.where(pre > 0, pre, 0.2 * pre))
    if pad:
        assert (Cd[:, N:] == 7.0).all() and (Pd[:, N:] == 0.0).all()
    C1 = torch.full((M, ldc), 7.0, device="cuda")                  # plain mode: bias + activation, no pre
    run(Ad, Bd, C1, M, N, K, lda, ldb, ldc, bias=dev(bias), act=2)
    close(C1[:, :N], np.maximum(pre, 0.0))
    if pad:
        assert (C1[:, N:] == 7.0).all()
    C2 = torch.ones(M, ldc, device="cuda")
    run(Ad, Bd, C2, M, N, K, lda, ldb, ldc, accumulate=True)
    close(C2[:, :N], A @ Bm + 1.0)
    C3 = torch.zeros(M, ldc, device="cuda")                        # same products as the 64x64 tiled kernel
    be.gemm_tile(Ad, Bd, C3, M, N, K, lda, ldb, ldc, 64, 64, bias=dev(bias))
    C4 = torch.zeros(M, ldc, device="cuda")
    run(Ad, Bd, C4, M, N, K, lda, ldb, ldc, bias=dev(bias))
    assert (C3[:, :N] - C4[:, :N]).abs().max().item() <= 2e-5 * max(1.0, float(np.abs(pre).max()))


@pytest.mark.parametrize("M,N,K,tA,tB,batch,cs,cfg", [
    (1024, 2048, 512, 0, 0, 1, 0, 0), (512, 2048, 1024, 1, 0, 2, 1, 0), (512, 5001, 960, 1, 0, 1, 1, 26),
    (960, 512, 256, 0, 1, 1, 0, 21), (37, 70, 50, 1, 0, 2, 1, 21), (130, 33, 77, 0, 0, 1, 0, 25), (65, 161, 40, 0, 1, 1, 0, 26),
    (100, 200, 96, 1, 0, 1, 1, 24), (960, 5001, 512, 0, 0, 1, 0, 28), (161, 129, 33, 0, 0, 1, 0, 1), (70, 70, 70, 1, 0, 1, 1, 9)])
def test_gemm_fused(be, M, N, K, tA, tB, batch, cs, cfg):
    """tnt_gemm_fused_f32 (one-round family: gemm1r / gemm2) on every layout, with ragged edges, the fused column sums
    of B (bias gradient) and the second product sharing B, against float64 numpy; padding columns hold garbage."""
    rng = np.random.default_rng(M + N + K)
    r4 = lambda n: (n + 3) // 4 * 4
    lda, ldb, ldc = r4(M if tA else K), r4(K if tB else N), r4(N) + 4
    A = rng.standard_normal(((K if tA else M), lda)); A2 = rng.standard_normal(A.shape)
    Bm = rng.standard_normal(((N if tB else K), ldb))
    opA = (A[:, :M].T if tA else A[:, :K]); opA2 = (A2[:, :M].T if tA else A2[:, :K])
    opB = (Bm[:, :K].T if tB else Bm[:, :N])
    bias = rng.standard_normal(N) if not cs else None
    C, C2 = torch.full((M, ldc), 5.0, device="cuda"), torch.full((M, ldc), 5.0, device="cuda")
    col = torch.full((ldc,), 5.0, device="cuda")
    be.gemm_fused(dev(A), dev(Bm), C, M, N, K, lda, ldb, ldc, transA=bool(tA), transB=bool(tB),
                  bias=dev(bias) if bias is not None else None, colsum=col if cs else None,
                  A2=dev(A2) if batch == 2 else None, C2=C2 if batch == 2 else None, cfg=cfg)
    close(C[:, :N], opA @ opB + (bias if bias is not None else 0))
    assert torch.equal(C[:, N:], torch.full((M, ldc - N), 5.0, device="cuda")), "padding columns written"
    if batch == 2:
        close(C2[:, :N], opA2 @ opB)
    if cs:
        close(col[:N], opB.sum(0), atol=1e-4 * np.abs(opB).sum(0).max())
        assert torch.equal(col[N:], torch.full((ldc - N,), 5.0, device="cuda"))


G3_TILES = list(range(1, 12))


def _g3_operands(M, N, K, tA, tB, seed):
    """zero-padded operands (the K-contiguous operand's pad columns must be zero: include/tnt_hip.h) + float64 views"""
    rng = np.random.default_rng(seed)
    r4 = lambda n: (n + 3) // 4 * 4
    lda, ldb, ldc = r4(M if tA else K), r4(K if tB else N), r4(N) + 4
    A = np.zeros(((K if tA else M), lda)); A[:, :(M if tA else K)] = rng.standard_normal(((K if tA else M), (M if tA else K)))
    A2 = np.zeros_like(A); A2[:, :(M if tA else K)] = rng.standard_normal(((K if tA else M), (M if tA else K)))
    Bm = np.zeros(((N if tB else K), ldb)); Bm[:, :(K if tB else N)] = rng.standard_normal(((N if tB else K), (K if tB else N)))
    opA, opA2 = (A[:, :M].T if tA else A[:, :K]), (A2[:, :M].T if tA else A2[:, :K])
    opB = (Bm[:, :K].T if tB else Bm[:, :N])
    return (A, A2, Bm, lda, ldb, ldc), (opA, opA2, opB)


@pytest.mark.parametrize("M,N,K,tA,tB", [
    (960, 5001, 512, 0, 0), (512, 5001, 960, 1, 0), (960, 512, 5001, 0, 1), (1024, 512, 2048, 0, 1), (512, 2048, 1024, 1, 0),
    (64, 501, 120, 1, 0), (120, 501, 64, 0, 0), (120, 64, 501, 0, 1), (37, 101, 50, 0, 0), (37, 101, 50, 1, 0), (37, 101, 50, 0, 1),
    (161, 129, 33, 0, 0), (5, 7, 3, 0, 1),
])
def test_gemm3(be, M, N, K, tA, tB):
    """tnt_gemm3_f32 (csrc/gemm3.hip: LDS-DMA staged FP32-MFMA family), EVERY tile configuration on every operand layout,
    ragged edges (rows / columns / K tails that end inside a tile, a 16-byte chunk, a K stage), with the bias rider,
    against float64 numpy; the padding columns of C stay untouched; results bitwise equal run to run."""
    (A, A2, Bm, lda, ldb, ldc), (opA, _, opB) = _g3_operands(M, N, K, tA, tB, M + N + K)
    bias = np.random.default_rng(1).standard_normal(N)
    want = opA @ opB + bias
    Ad, Bd, bd = dev(A), dev(Bm), dev(bias)
    tol = 1e-6 * max(1.0, (K / 256) ** 0.5)
    for tile in G3_TILES:
        outs = []
        for rep in range(3):
            C = torch.full((M, ldc), 5.0, device="cuda")
            be.gemm3(Ad, Bd, C, M, N, K, lda, ldb, ldc, transA=bool(tA), transB=bool(tB), bias=bd, tile=tile)
            outs.append(C)
        close(outs[0][:, :N], want, rtol=tol)
        assert torch.equal(outs[0][:, N:], torch.full((M, ldc - N), 5.0, device="cuda")), ("padding columns written", tile)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), ("not reproducible", tile)


@pytest.mark.parametrize("M,N,K,tA,tB,S", [
    (960, 512, 5001, 0, 1, 2), (960, 512, 5001, 0, 1, 4), (1024, 512, 2048, 0, 1, 2), (256, 5001, 960, 1, 0, 2), (960, 2048, 544, 0, 0, 3),
    (64, 501, 120, 1, 0, 2), (37, 101, 150, 0, 1, 3), (161, 129, 257, 0, 0, 8), (100, 60, 1000, 1, 0, 6),
])
def test_gemm3_splitk_in_launch(be, M, N, K, tA, tB, S):
    """K split over S workgroups per tile with the reduction INSIDE the launch (the sentinel-armed exchange buffer of
    csrc/gemm3.hip): exact result against float64, bitwise reproducible over launches that reuse ONE work buffer, the buffer
    fully re-armed after every launch (so launches of other shapes can share it), the error word untouched."""
    (A, _, Bm, lda, ldb, ldc), (opA, _, opB) = _g3_operands(M, N, K, tA, tB, M * 3 + N + K)
    want = opA @ opB
    Ad, Bd = dev(A), dev(Bm)
    tol = 1e-6 * max(1.0, (K / 256) ** 0.5)
    for tile in (1, 3, 4, 5, 7, 9):
        wf = be.gemm3_work_floats(M, N, tile, S)
        if wf <= 0:
            continue
        work = torch.empty((wf + 3) // 4 * 4, device="cuda")
        be.gemm3_work_arm(work)
        armed = work.view(torch.int32).clone()
        assert bool((armed == 0x7FC5EED5).all())
        sync = torch.zeros(be.gemm3_sync_words(M, N, tile), dtype=torch.int32, device="cuda")
        outs = []
        for rep in range(3):
            C = torch.full((M, ldc), 5.0, device="cuda")
            try:
                be.gemm3(Ad, Bd, C, M, N, K, lda, ldb, ldc, transA=bool(tA), transB=bool(tB), tile=tile, splitk=S, work=work, sync=sync)
            except RuntimeError:          # more workgroups than one round holds: refused (TNT_BADARG), never a hang
                outs = None
                break
            outs.append(C)
            torch.cuda.synchronize()
            assert torch.equal(work.view(torch.int32), armed), ("exchange buffer not re-armed", tile, rep)
        if outs is None:
            continue
        assert int(sync.sum()) == 0, ("error word set", tile)
        close(outs[0][:, :N], want, rtol=tol)
        assert torch.equal(outs[0][:, N:], torch.full((M, ldc - N), 5.0, device="cuda")), ("padding columns written", tile)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), ("not reproducible", tile)


@pytest.mark.parametrize("M,N,K", [(512, 2048, 1024), (512, 5001, 960), (64, 256, 128), (64, 501, 120), (37, 101, 50), (130, 70, 333)])
def test_gemm3_riders(be, M, N, K):
    """the TN riders of tnt_gemm3_f32: column sums of B (the bias gradient of the layer whose kernel gradient the product
    is) and a second product sharing B (LSTM kernel + recurrent-kernel gradients, NIC.py:138-140 under tape.gradient) in
    ONE launch -- each output against float64, padding untouched, bitwise reproducible."""
    (A, A2, Bm, lda, ldb, ldc), (opA, opA2, opB) = _g3_operands(M, N, K, 1, 0, M + 2 * N + K)
    Ad, A2d, Bd = dev(A), dev(A2), dev(Bm)
    tol = 1e-6 * max(1.0, (K / 256) ** 0.5)
    for tile in G3_TILES:
        outs = []
        for rep in range(2):
            C, C2 = torch.full((M, ldc), 5.0, device="cuda"), torch.full((M, ldc), 5.0, device="cuda")
            col = torch.full((ldc,), 5.0, device="cuda")
            be.gemm3(Ad, Bd, C, M, N, K, lda, ldb, ldc, transA=True, colsum=col, A2=A2d, C2=C2, tile=tile)
            outs.append((C, C2, col))
        C, C2, col = outs[0]
        close(C[:, :N], opA @ opB, rtol=tol)
        close(C2[:, :N], opA2 @ opB, rtol=tol)
        close(col[:N], opB.sum(0), atol=2e-6 * np.abs(opB).sum(0).max())
        for x in (C, C2):
            assert torch.equal(x[:, N:], torch.full((M, ldc - N), 5.0, device="cuda")), tile
        assert torch.equal(col[N:], torch.full((ldc - N,), 5.0, device="cuda"))
        assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), ("not reproducible", tile)


@pytest.mark.parametrize("shape1,shape2,t1,t2,S2", [
    ((512, 5001, 960), (960, 512, 5001), 4, 7, 2), ((512, 2048, 1024), (1024, 512, 2048), 5, 7, 2), ((256, 5001, 960), (960, 256, 5001), 4, 7, 4),
    ((64, 501, 120), (120, 64, 501), 4, 7, 1), ((64, 256, 128), (128, 64, 256), 5, 5, 2), ((37, 101, 50), (50, 37, 101), 5, 7, 1),
    ((512, 256, 960), (960, 512, 256), 7, 7, 1), ((512, 256, 960), (960, 512, 256), 7, 5, 2),
])
def test_gemm3_pair(be, shape1, shape2, t1, t2, S2):
    """tnt_gemm3_pair_f32: a TN product (second product + column sums riding along) and an independent NT product (K split
    in the launch) as ONE grid -- each output bit-identical to the same product launched on its own."""
    (M1, N1, K1), (M2, N2, K2) = shape1, shape2
    (A, A2, Bm, lda, ldb, ldc), _ = _g3_operands(M1, N1, K1, 1, 0, 3)
    (P, _, Q, ldp, ldq, ldr), (opP, _, opQ) = _g3_operands(M2, N2, K2, 0, 1, 4)
    Ad, A2d, Bd, Pd, Qd = dev(A), dev(A2), dev(Bm), dev(P), dev(Q)
    wf = (be.gemm3_work_floats(M2, N2, t2, S2) + 3) // 4 * 4 if S2 > 1 else 0
    work = torch.empty(max(wf, 4), device="cuda")
    be.gemm3_work_arm(work)
    sync = torch.zeros(1, dtype=torch.int32, device="cuda")
    if S2 > 1 and wf == 0:
        pytest.skip("no split for this shape")
    # separately
    C, C2, col = (torch.full((M1, ldc), 5.0, device="cuda") for _ in range(2)) , None, None
    C, C2 = C
    col = torch.full((ldc,), 5.0, device="cuda")
    R = torch.full((M2, ldr), 5.0, device="cuda")
    be.gemm3(Ad, Bd, C, M1, N1, K1, lda, ldb, ldc, transA=True, colsum=col, A2=A2d, C2=C2, tile=t1)
    be.gemm3(Pd, Qd, R, M2, N2, K2, ldp, ldq, ldr, transB=True, tile=t2, splitk=S2, work=work if S2 > 1 else None,
             sync=sync if S2 > 1 else None)
    close(R[:, :N2], opP @ opQ, rtol=1e-6 * max(1.0, (K2 / 256) ** 0.5))
    # as a pair
    assert be.gemm3_pair_supported(t1, True, False, t2, False, True)
    for rep in range(2):
        Cp, C2p, Rp = torch.full((M1, ldc), 5.0, device="cuda"), torch.full((M1, ldc), 5.0, device="cuda"), torch.full((M2, ldr), 5.0, device="cuda")
        colp = torch.full((ldc,), 5.0, device="cuda")
        d1 = be.gemm3_desc(Ad, Bd, Cp, M1, N1, K1, lda, ldb, ldc, transA=True, colsum=colp, A2=A2d, C2=C2p, tile=t1)
        d2 = be.gemm3_desc(Pd, Qd, Rp, M2, N2, K2, ldp, ldq, ldr, transB=True, tile=t2, splitk=S2, work=work if S2 > 1 else None,
                           sync=sync if S2 > 1 else None)
        be.gemm3_pair(d1, d2)
        torch.cuda.synchronize()
        assert torch.equal(Cp, C) and torch.equal(C2p, C2) and torch.equal(colp, col) and torch.equal(Rp, R), rep
    assert int(sync.sum()) == 0 and bool((work.view(torch.int32) == 0x7FC5EED5).all())
    assert not be.gemm3_pair_supported(1, False, False, 7, False, True)


@pytest.mark.parametrize("shape1,shape2,t1,t2,S1,S2", [((32, 2048, 960), (512, 32, 960), 5, 7, 4, 8), ((32, 2048, 960), (512, 32, 960), 7, 7, 1, 1),
                                                        ((64, 64, 100), (20, 40, 30), 7, 5, 1, 1), ((96, 256, 512), (64, 192, 640), 5, 5, 2, 2)])
def test_gemm3_pair_of_two_kernel_gradients(be, shape1, shape2, t1, t2, S1, S2):
    """tnt_gemm3_pair_f32 with TWO TN products (two small kernel gradients of one step; each may split K inside the launch, on
    its own exchange space): every output bit-identical to the same product launched on its own."""
    outs = []
    sync = torch.zeros(1, dtype=torch.int32, device="cuda")
    specs = []
    for (M, N, K), t, S, seed in ((shape1, t1, S1, 3), (shape2, t2, S2, 4)):
        (A, _, Bm, lda, ldb, ldc), (opA, _, opB) = _g3_operands(M, N, K, 1, 0, seed)
        wf = (be.gemm3_work_floats(M, N, t, S) + 3) // 4 * 4 if S > 1 else 0
        if S > 1 and wf == 0:
            pytest.skip("no split for this shape")
        work = torch.empty(max(wf, 4), device="cuda")
        be.gemm3_work_arm(work)
        Ad, Bd = dev(A), dev(Bm)
        C = torch.full((M, ldc), 5.0, device="cuda")
        be.gemm3(Ad, Bd, C, M, N, K, lda, ldb, ldc, transA=True, tile=t, splitk=S, work=work if S > 1 else None, sync=sync if S > 1 else None)
        close(C[:, :N], opA @ opB, rtol=1e-6 * max(1.0, (K / 256) ** 0.5))
        specs.append((Ad, Bd, C, M, N, K, lda, ldb, ldc, t, S, work))
    assert be.gemm3_pair_supported(t1, True, False, t2, True, False)
    for rep in range(2):
        ds, got = [], []
        for Ad, Bd, C, M, N, K, lda, ldb, ldc, t, S, work in specs:
            Cp = torch.full((M, ldc), 5.0, device="cuda")
            got.append(Cp)
            ds.append(be.gemm3_desc(Ad, Bd, Cp, M, N, K, lda, ldb, ldc, transA=True, tile=t, splitk=S, work=work if S > 1 else None,
                                    sync=sync if S > 1 else None))
        be.gemm3_pair(ds[0], ds[1])
        torch.cuda.synchronize()
        assert torch.equal(got[0], specs[0][2]) and torch.equal(got[1], specs[1][2]), rep
    assert int(sync.sum()) == 0
    assert not be.gemm3_pair_supported(4, True, False, 7, True, False)


def test_gemm3_plan_and_bad_arguments(be):
    """the cost model's pick runs and is exact on every hot-path shape of BASELINE configs 2 and 3; unsupported argument
    combinations are refused with an error code, never launched"""
    shapes = [(960, 5001, 512, 0, 0), (512, 5001, 960, 1, 0), (960, 512, 5001, 0, 1), (1024, 2048, 512, 0, 0), (1024, 512, 2048, 0, 1),
              (960, 5001, 256, 0, 0), (256, 5001, 960, 1, 0), (960, 256, 5001, 0, 1), (960, 2048, 544, 0, 0), (960, 544, 2048, 0, 1),
              (32, 2048, 960, 1, 0), (512, 256, 960, 1, 0), (960, 512, 256, 0, 1)]
    for M, N, K, tA, tB in shapes:
        (A, _, Bm, lda, ldb, ldc), (opA, _, opB) = _g3_operands(M, N, K, tA, tB, 7)
        tile, S = be.gemm3_plan(M, N, K, bool(tA), bool(tB))
        assert 1 <= tile <= 11 and 1 <= S <= 8
        wf = be.gemm3_work_floats(M, N, tile, S)
        work = torch.empty((max(wf, 4) + 3) // 4 * 4, device="cuda")
        be.gemm3_work_arm(work)
        sync = torch.zeros(1, dtype=torch.int32, device="cuda")
        C = torch.zeros(M, ldc, device="cuda")
        be.gemm3(dev(A), dev(Bm), C, M, N, K, lda, ldb, ldc, transA=bool(tA), transB=bool(tB), tile=tile, splitk=S, work=work, sync=sync)
        close(C[:, :N], opA @ opB, rtol=1e-6 * max(1.0, (K / 256) ** 0.5))
        assert int(sync.sum()) == 0
    A = torch.zeros(64, 64, device="cuda")
    with pytest.raises(RuntimeError):
        be.gemm3(A, A, A, 64, 64, 64, 64, 64, 64, transA=True, transB=True)            # both transposed
    with pytest.raises(RuntimeError):
        be.gemm3(A, A, A, 64, 64, 64, 62, 64, 64)                                        # leading dimension not a multiple of 4
    with pytest.raises(RuntimeError):
        be.gemm3(A, A, A, 64, 64, 64, 64, 64, 64, splitk=2)                              # split without work / sync
    with pytest.raises(RuntimeError):
        be.gemm3(A, A, A, 64, 64, 64, 64, 64, 64, colsum=A)                              # column sums need transA
    with pytest.raises(RuntimeError):
        be.gemm3(A, A, A, 64, 64, 64, 64, 64, 64, tile=99)


def test_gemm_splitk(be):
    rng = np.random.default_rng(0)
    M, N, K = 64, 512, 20000
    A, Bm, bias = rng.standard_normal((M, K)), rng.standard_normal((K, N)), rng.standard_normal(N)
    splitk = 40
    work = torch.empty(splitk * M * N, device="cuda")
    Cd = torch.zeros(M, N, device="cuda"); Pd = torch.zeros(M, N, device="cuda")
    be.gemm(dev(A), dev(Bm), Cd, M, N, K, K, N, N, bias=dev(bias), pre=Pd, act=1, splitk=splitk, work=work)
    pre = A @ Bm + bias
    close(Pd, pre)
    close(Cd, np.where(pre > 0, pre, 0.2 * pre))


def test_gemm_graph_capture(be):
    """The library's launches must be capturable on torch's stream (hipGraph replay)."""
    rng = np.random.default_rng(1)
    A, Bm = rng.standard_normal((128, 64)), rng.standard_normal((64, 128))
    Ad, Bd, Cd = dev(A), dev(Bm), torch.zeros(128, 128, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        be.gemm(Ad, Bd, Cd, 128, 128, 64, 64, 128, 128)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        be.gemm(Ad, Bd, Cd, 128, 128, 64, 64, 128, 128)
    Cd.zero_()
    Ad.copy_(dev(2 * A))
    g.replay()
    torch.cuda.synchronize()
    close(Cd, 2 * A @ Bm)


# --------------------------------------------------------------------------- dropout
@pytest.mark.parametrize("tmajor", [False, True])
def test_dropout_bit_exact(be, tmajor):
    rng = np.random.default_rng(2)
    B, T, E, rate = 6, 5, 37, 0.2
    x = rng.standard_normal((B, T, E)).astype(np.float32)
    keep = keep_mask((B, T, E + 3), rate, seed=1234567890123, site=49, step=7)[:, :, 3:]   # lcol0 = 3
    want = np.where(keep, x * np.float32(1.0 / (1.0 - np.float32(rate))), 0).astype(np.float32)
    step_dev = torch.tensor([4], dtype=torch.int32, device="cuda")
    if tmajor:
        xd = dev(x.transpose(1, 0, 2).reshape(T * B, E)); yd = torch.zeros_like(xd)
        be.dropout(xd, yd, T * B, E, E, B, E + 3, 3, rate, 1234567890123, 49, 3, step_dev)
        got = yd.cpu().numpy().reshape(T, B, E).transpose(1, 0, 2)
    else:
        xd = dev(x.reshape(B * T, E)); yd = torch.zeros_like(xd)
        be.dropout(xd, yd, B * T, E, E, 0, E + 3, 3, rate, 1234567890123, 49, 3, step_dev)
        got = yd.cpu().numpy().reshape(B, T, E)
    assert np.array_equal(got, want)


def test_dropout_rows_per_site(be):
    rng = np.random.default_rng(21)
    T, B, E, D, rate = 5, 6, 8, 4, 0.3
    x = rng.standard_normal((T * B, E)).astype(np.float32)
    xd, yd = dev(x), torch.zeros(T * B, E, device="cuda")
    be.dropout(xd, yd, T * B, E, E, 0, D + E, D, rate, 99, 48, 2, None, rows_per_site=B)
    scale = np.float32(1.0 / (1.0 - np.float32(rate)))
    for i in range(T):
        keep = keep_mask((B, D + E), rate, 99, 48 + i, 2)[:, D:]
        want = np.where(keep, x[i * B:(i + 1) * B] * scale, 0).astype(np.float32)
        assert np.array_equal(yd[i * B:(i + 1) * B].cpu().numpy(), want)


def test_two_masks_in_one_pass(be):
    """tnt_dropout2_f32 and tnt_embedding_fwd_drop2_f32 (the Embedding Dropout and the per-timestep LSTM input mask of the
    text rows, lc_NIC.py:233-234,255) give the bits of the two launches they replace, and the oracle's Philox pattern."""
    rng = np.random.default_rng(5)
    T, B, E, D, V, r_text, r_lstm, seed, s_text, s_in, step = 7, 6, 16, 8, 50, 0.25, 0.4, 31337, 7, 48, 3
    step_dev = torch.tensor([step], dtype=torch.int32, device="cuda")
    x = dev(rng.standard_normal((T * B, E)))
    want = x.clone()
    be.dropout(want, want, T * B, E, E, 0, D + E, D, r_lstm, seed, s_in, 0, step_dev, rows_per_site=B)
    be.dropout(want, want, T * B, E, E, B, E, 0, r_text, seed, s_text, 0, step_dev)
    got = torch.zeros_like(x)
    be.dropout2(x, got, T * B, E, E, (0, D + E, D, B, r_lstm, s_in), (B, E, 0, 0, r_text, s_text), seed, 0, step_dev)
    assert torch.equal(got, want)
    k_text = keep_mask((B, T, E), r_text, seed, s_text, step).transpose(1, 0, 2)                    # (T, B, E)
    k_in = np.stack([keep_mask((B, D + E), r_lstm, seed, s_in + t, step)[:, D:] for t in range(T)])
    assert np.array_equal(got.cpu().numpy().reshape(T, B, E) != 0, k_text & k_in & (x.cpu().numpy().reshape(T, B, E) != 0))
    table, ids = dev(rng.standard_normal((V, E))), torch.tensor(rng.integers(0, V, (B, T)), dtype=torch.int32, device="cuda")
    w2 = torch.zeros(T * B, E, device="cuda")
    be.embedding_fwd_drop(table, ids, None, w2, B, T, E, E, V, r_text, seed, s_text, 0, step_dev)
    be.dropout(w2, w2, T * B, E, E, 0, D + E, D, r_lstm, seed, s_in, 0, step_dev, rows_per_site=B)
    g2 = torch.zeros(T * B, E, device="cuda")
    be.embedding_fwd_drop(table, ids, None, g2, B, T, E, E, V, r_text, seed, s_text, 0, step_dev, mask2=(r_lstm, s_in, D + E, D))
    assert torch.equal(g2, w2)
    assert np.array_equal(g2.cpu().numpy().reshape(T, B, E) != 0, k_text & k_in)


# ------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("rows,C", [(64, 512), (1440, 32), (7, 5)])
def test_batchnorm(be, rows, C):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((rows, C)) * 2 + 1.5
    gam, bet = rng.standard_normal(C), rng.standard_normal(C)
    mm0, mv0 = rng.standard_normal(C), rng.random(C) + 0.5
    dy = rng.standard_normal((rows, C))
    nch = be.bn_nchunk(rows)
    work = torch.zeros(C * (2 * nch + 1), device="cuda")
    for training in (True, False):
        mm, mv = dev(mm0), dev(mv0)
        y, xhat, inv = torch.zeros(rows, C, device="cuda"), torch.zeros(rows, C, device="cuda"), torch.zeros(C, device="cuda")
        be.batchnorm_fwd(dev(x), dev(gam), dev(bet), mm, mv, y, xhat, inv, rows, C, C, training, 1e-3, 0.99, work)
        yo, cache, mmo, mvo = O.batchnorm_fwd(x, gam, bet, mm0, mv0, training)
        close(y, yo); close(mm, mmo); close(mv, mvo)
        dx, dg, db = torch.zeros(rows, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        be.batchnorm_bwd(dev(dy), xhat, dev(gam), inv, dx, dg, db, rows, C, C, training, work)
        dxo, dgo, dbo = O.batchnorm_bwd(dy, gam, cache)
        close(dx, dxo); close(dg, dgo); close(db, dbo)


@pytest.mark.parametrize("rows,C", [(23040, 32), (130, 36), (7, 5)])
def test_batchnorm_with_dropout_and_activation_gradient(be, rows, C):
    """tnt_batchnorm_fwd_drop_f32: y = Dropout(BN(x)) with the Philox pattern of tnt_dropout_f32 (bit pattern from the
    oracle's stream), xhat un-dropped; tnt_batchnorm_bwd_act_f32: dx = BN'(dy) * LeakyReLU'(pre)."""
    rng = np.random.default_rng(rows)
    x, dy, pre = rng.standard_normal((rows, C)) * 2 + 0.5, rng.standard_normal((rows, C)), rng.standard_normal((rows, C))
    gam, bet, mm0, mv0 = rng.standard_normal(C), rng.standard_normal(C), rng.standard_normal(C), rng.random(C) + 0.5
    rate, seed, site, step = 0.3, 987654321, 21, 6
    step_dev = torch.tensor([step], dtype=torch.int32, device="cuda")
    work = torch.zeros(C * (2 * be.bn_nchunk(rows) + 1), device="cuda")
    mm, mv = dev(mm0), dev(mv0)
    y, xhat, inv = torch.zeros(rows, C, device="cuda"), torch.zeros(rows, C, device="cuda"), torch.zeros(C, device="cuda")
    be.batchnorm_fwd(dev(x), dev(gam), dev(bet), mm, mv, y, xhat, inv, rows, C, C, True, 1e-3, 0.99, work,
                     drop=(rate, seed, site, step_dev))
    yo, cache, mmo, mvo = O.batchnorm_fwd(x, gam, bet, mm0, mv0, True)
    keep = keep_mask((rows, C), rate, seed, site, step)
    close(y, O.dropout_fwd(yo, keep, rate)); close(xhat, cache[0]); close(mm, mmo); close(mv, mvo)
    assert np.array_equal(y.cpu().numpy() == 0, ~keep | (yo.astype(np.float32) == 0))
    # the two-launch form gives the same bits
    y2, xh2, inv2 = torch.zeros_like(y), torch.zeros_like(xhat), torch.zeros_like(inv)
    be.batchnorm_fwd(dev(x), dev(gam), dev(bet), dev(mm0), dev(mv0), y2, xh2, inv2, rows, C, C, True, 1e-3, 0.99, work)
    be.dropout(y2, y2, rows, C, C, 0, C, 0, rate, seed, site, 0, step_dev)
    assert torch.equal(y, y2) and torch.equal(xhat, xh2)
    dx, dg, db = torch.zeros(rows, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    be.batchnorm_bwd(dev(dy), xhat, dev(gam), inv, dx, dg, db, rows, C, C, True, work, act_pre=dev(pre), slope=0.2)
    dxo, dgo, dbo = O.batchnorm_bwd(dy, gam, cache)
    close(dx, dxo * np.where(pre > 0, 1.0, 0.2)); close(dg, dgo); close(db, dbo)


@pytest.mark.parametrize("rows,C", [(64, 512), (9, 33)])
def test_layernorm(be, rows, C):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((rows, C)) * 2 + 1.5
    gam, bet, dy = rng.standard_normal(C), rng.standard_normal(C), rng.standard_normal((rows, C))
    y, xhat, inv = torch.zeros(rows, C, device="cuda"), torch.zeros(rows, C, device="cuda"), torch.zeros(rows, device="cuda")
    be.layernorm_fwd(dev(x), dev(gam), dev(bet), y, xhat, inv, rows, C, C, 1e-3)
    yo, cache = O.layernorm_fwd(x, gam, bet)
    close(y, yo)
    work = torch.zeros(C * (2 * be.bn_nchunk(rows) + 1), device="cuda")
    dx, dg, db = torch.zeros(rows, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    be.layernorm_bwd(dev(dy), xhat, dev(gam), inv, dx, dg, db, rows, C, C, work)
    dxo, dgo, dbo = O.layernorm_bwd(dy, gam, cache)
    close(dx, dxo); close(dg, dgo); close(db, dbo)


def test_colsum_sum_actbwd(be):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((960, 5001))
    xd = torch.zeros(960, 5004, device="cuda"); xd[:, :5001] = dev(x)
    out = torch.zeros(5001, device="cuda")
    work = torch.zeros(5001 * be.bn_nchunk(960), device="cuda")
    be.colsum(xd, out, 960, 5001, 5004, work)
    close(out, x.sum(0))
    v = rng.standard_normal(960)
    o = torch.zeros(1, device="cuda")
    be.sum(dev(v), o, 960, 0.5)
    close(o, [0.5 * v.sum()])
    pre, dy = rng.standard_normal(1000), rng.standard_normal(1000)
    dx = torch.zeros(1000, device="cuda")
    be.act_bwd(dev(pre), dev(dy), dx, 1000, 1, 0.2)
    close(dx, np.where(pre > 0, dy, 0.2 * dy))


@pytest.mark.parametrize("rows,C,pad", [(1, 1, 0), (15, 64, 0), (64, 33, 1), (960, 2048, 0), (2048, 37, 3), (2049, 37, 3),
                                        (23040, 32, 0)])
def test_colsum_short_and_long(be, rows, C, pad):
    """tnt_colsum_f32: one direct launch up to 2048 rows, chunk partials + finalize beyond; padding columns ignored."""
    rng = np.random.default_rng(rows * 3 + C)
    x = rng.standard_normal((rows, C))
    xd = torch.full((rows, C + pad), 1e6, device="cuda"); xd[:, :C] = dev(x)
    out = torch.full((C + 2,), -5.0, device="cuda")
    work = torch.zeros(max(1, C * be.bn_nchunk(rows)), device="cuda")
    be.colsum(xd, out, rows, C, C + pad, work)
    close(out[:C], x.sum(0))
    assert (out[C:] == -5.0).all()


# -------------------------------------------------------------------------- embedding
@pytest.mark.parametrize("B,T,E,V", [(8, 5, 70, 23), (64, 15, 512, 5001), (9, 4, 36, 11)])
def test_embedding(be, B, T, E, V):
    rng = np.random.default_rng(6)
    table = rng.standard_normal((V, E))
    ids = rng.integers(0, V, (B, T)).astype(np.int32)
    ids[0, :] = 3                                        # duplicates
    ids[:, T // 2:] = np.where(rng.random((B, T - T // 2)) < 0.7, 0, ids[:, T // 2:])   # heavy pad token
    out = torch.zeros(T * B, E, device="cuda")
    idd = dev(ids, torch.int32)
    be.embedding_fwd(dev(table), idd, out, B, T, E, E, V)
    want = table[ids].transpose(1, 0, 2).reshape(T * B, E)
    assert np.array_equal(out.cpu().numpy(), want.astype(np.float32))
    drows = rng.standard_normal((B, T, E))
    dtab = torch.full((V, E), 9.0, device="cuda")
    sq = torch.zeros(1, device="cuda")
    be.embedding_bwd(dev(drows.transpose(1, 0, 2).reshape(T * B, E)), idd, dtab, sq, torch.zeros(B * T, device="cuda"),
                     B, T, E, E, V)
    close(dtab, O.embedding_bwd_dense(drows, ids, V))
    close(sq, [(drows ** 2).sum()])


# ------------------------------------------------------------------------------- LSTM
@pytest.mark.parametrize("B,T,E,V", [(64, 15, 512, 5001), (8, 5, 72, 23), (9, 4, 36, 11)])
def test_embedding_bwd_sparse_over_consecutive_steps(be, B, T, E, V):
    """tnt_embedding_bwd_sparse_f32 over four consecutive "steps" with different token ids (heavy duplicates, ids that
    disappear again): after every step the gradient table equals the dense scatter of THAT step alone -- rows only the
    previous step touched are cleaned, nothing else is written -- and the norm partials add up to the squared norm of
    the un-deduplicated rows; tnt_step_finalize_f32 hands the ids on and sums the partials."""
    rng = np.random.default_rng(B * T)
    n = B * T
    table = torch.zeros(V, E, device="cuda")
    prev = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    nparts = be.embedding_bwd_parts(B, T, E)
    parts, sq = torch.zeros(nparts, device="cuda"), torch.zeros(1, device="cuda")
    none = torch.zeros(1, device="cuda")
    for step in range(4):
        ids = rng.integers(0, min(V, 6 + 3 * step), (B, T)).astype(np.int32)
        ids[:, T // 2:] = rng.integers(0, V, (B, T - T // 2))
        if step == 2:
            ids[:] = 1                                          # one id owns every row
        drows = rng.standard_normal((T, B, E))
        idd = dev(ids, torch.int32)
        if step % 2 == 1 and E % 4 == 0:
            # with the forward's input-dropout mask folded in: same result as a dropout launch over the rows in front
            step_dev = torch.tensor([3 + step], dtype=torch.int32, device="cuda")
            dd = dev(drows.reshape(T * B, E)).clone()
            be.dropout(dd, dd, T * B, E, E, B, E, 0, 0.3, 91, 49, 0, step_dev)
            want_rows = dd.cpu().double().numpy().reshape(T, B, E)
            be.embedding_bwd_sparse(dev(drows.reshape(T * B, E)), idd, prev, table, parts, B, T, E, E, V, drop_rate=0.3,
                                    drop_seed=91, drop_site=49, drop_step_dev=step_dev)
            drows = want_rows
        else:
            be.embedding_bwd_sparse(dev(drows.reshape(T * B, E)), idd, prev, table, parts, B, T, E, E, V)
        want = O.embedding_bwd_dense(np.transpose(drows, (1, 0, 2)), ids, V)
        be.step_finalize(none, None, None, None, None, None, 0, extra_part=parts, extra=sq, n_extra=nparts, ids_src=idd,
                         ids_dst=prev, n_ids=n)
        torch.cuda.synchronize()
        close(table, want, atol=1e-5 * np.abs(want).max())
        assert torch.equal(table == 0, torch.tensor(want == 0, device="cuda")), "a row outside this step's ids is not zero"
        assert abs(float(sq) - (drows ** 2).sum()) <= 1e-5 * (drows ** 2).sum()
        assert torch.equal(prev, idd.reshape(-1))


def test_embedding_bwd_sparse_skips_the_rows_of_the_mask_id(be):
    """zero_id: rows of that id are declared zero by the caller and never read -- poisoned here with NaN to show it; the table
    equals the dense scatter with those rows at zero, the mask id's own row is written as zeros (also when a previous step
    left something there), the norm is that of the other rows."""
    rng = np.random.default_rng(5)
    B, T, E, V = 64, 15, 512, 5001
    n = B * T
    table = torch.full((V, E), 3.0, device="cuda")
    table[1:] = 0
    prev = torch.full((n,), -1, dtype=torch.int32, device="cuda"); prev[0] = 0
    nparts = be.embedding_bwd_parts(B, T, E)
    parts, sq, none = torch.zeros(nparts, device="cuda"), torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    ids = rng.integers(1, 40, (B, T)).astype(np.int32)
    for b in range(B):
        ids[b, rng.integers(3, T):] = 0                         # padded tails
    drows = rng.standard_normal((T, B, E))
    clean = drows.copy(); clean[(ids == 0).T] = 0.0
    drows[(ids == 0).T] = np.nan
    idd = dev(ids, torch.int32)
    be.embedding_bwd_sparse(dev(drows.reshape(T * B, E)), idd, prev, table, parts, B, T, E, E, V, zero_id=0)
    be.step_finalize(none, None, None, None, None, None, 0, extra_part=parts, extra=sq, n_extra=nparts, ids_src=idd,
                     ids_dst=prev, n_ids=n)
    torch.cuda.synchronize()
    want = O.embedding_bwd_dense(np.transpose(clean, (1, 0, 2)), ids, V)
    assert not torch.isnan(table).any()
    close(table, want, atol=1e-5 * np.abs(want).max())
    assert float(table[0].abs().max()) == 0.0
    assert abs(float(sq) - (clean ** 2).sum()) <= 1e-5 * (clean ** 2).sum()


def test_step_finalize(be):
    """tnt_step_finalize_f32 == seg_finalize + l2_total + sum2 + step_tick of the unfused sequence."""
    rng = np.random.default_rng(77)
    nseg, nspan = 37, 200
    first = np.sort(np.concatenate([[0, nspan], rng.choice(np.arange(1, nspan), nseg - 1, replace=False)])).astype(np.int32)
    partial = rng.random(2 * nspan)
    lam = rng.random(nseg) * 0.01
    x0, x1 = rng.standard_normal(960), rng.random(960)
    sq, wsq, l2, o0, o1 = (torch.zeros(k, device="cuda") for k in (nseg, nseg, 1, 1, 1))
    adam_t = torch.tensor([4], dtype=torch.int64, device="cuda")
    drop = torch.tensor([9], dtype=torch.int32, device="cuda")
    lr, lr_t = torch.tensor([1e-3], device="cuda"), torch.zeros(1, device="cuda")
    guard = torch.zeros(1, dtype=torch.int32, device="cuda")
    args = lambda: be.step_finalize(dev(partial), dev(first, torch.int32), dev(lam), sq, wsq, l2, nseg, x0=dev(x0), out0=o0,
                                    x1=dev(x1), out1=o1, n=960, scale=1 / 960, adam_t=adam_t, drop_step=drop, lr=lr, lr_t=lr_t,
                                    beta1=0.9, beta2=0.98, guard=guard)
    args()
    torch.cuda.synchronize()
    q = np.array([partial[2 * first[s]:2 * first[s + 1]:2].sum() for s in range(nseg)])
    w = np.array([partial[2 * first[s] + 1:2 * first[s + 1]:2].sum() for s in range(nseg)])
    close(sq, q); close(wsq, w); close(l2, [(lam * w).sum()]); close(o0, [x0.mean()], atol=1e-6); close(o1, [x1.mean()])
    assert int(adam_t) == 5 and int(drop) == 10
    assert abs(float(lr_t) - 1e-3 * np.sqrt(1 - 0.98 ** 5) / (1 - 0.9 ** 5)) < 1e-9
    guard[0] = 1                                               # a tripped device guard freezes the step state
    args()
    torch.cuda.synchronize()
    assert int(adam_t) == 5 and int(drop) == 10


@pytest.mark.parametrize("B,U", [(64, 512), (5, 16), (7, 1040)])
def test_ln_lstm_cell(be, B, U):
    """tnt_ln_lstm_cell_fwd/bwd_f32 + the two 4U-wide tnt_layernorm launches against oracle ln_lstm_step_fwd/bwd
    (tensorflow_addons LayerNormLSTMCell, lc_NIC.py:126-136): states, gates, every input gradient."""
    rng = np.random.default_rng(B + U)
    Din = 24
    x, h, c = rng.standard_normal((B, Din)), rng.standard_normal((B, U)) * 0.5, rng.standard_normal((B, U)) * 0.5
    W, Ur = rng.standard_normal((Din, 4 * U)) / np.sqrt(Din), rng.standard_normal((U, 4 * U)) / np.sqrt(U)
    b = rng.standard_normal(4 * U) * 0.1
    gk, bk, gr, br = (1 + 0.1 * rng.standard_normal(4 * U), 0.1 * rng.standard_normal(4 * U),
                      1 + 0.1 * rng.standard_normal(4 * U), 0.1 * rng.standard_normal(4 * U))
    gs, bs = 1 + 0.1 * rng.standard_normal(U), 0.1 * rng.standard_normal(U)
    h2, c2, cache = O.ln_lstm_step_fwd(x, h, c, W, Ur, b, gk, bk, gr, br, gs, bs)
    f = lambda *sh: torch.zeros(*sh, device="cuda")
    zk, zr, xhk, xhr, isk, isr = f(B, U, 4), f(B, U, 4), f(B, 4 * U), f(B, 4 * U), f(max(B, 4 * U)), f(max(B, 4 * U))
    be.layernorm_fwd(dev(il(x @ W, U)), dev(il(gk, U)), dev(il(bk, U)), zk, xhk, isk, B, 4 * U, 4 * U, 1e-3)
    be.layernorm_fwd(dev(il(h @ Ur, U)), dev(il(gr, U)), dev(il(br, U)), zr, xhr, isr, B, 4 * U, 4 * U, 1e-3)
    gates, chat, iss, cn, hn = f(B, U, 4), f(B, U), f(B), f(B, U), f(B, U)
    be.ln_lstm_cell_fwd(zk, zr, dev(il(b, U)), dev(c), dev(gs), dev(bs), gates, chat, iss, cn, hn, B, U, 1e-3)
    close(hn, h2); close(cn, c2)
    close(gates, np.stack(cache[3:7], axis=-1))
    dh2, dcn = rng.standard_normal((B, U)), rng.standard_normal((B, U))
    dx, dh, dc, gl = O.ln_lstm_step_bwd(dh2, dcn, cache, W, Ur, gk, gr, gs)
    dz, dcp, dcnt = f(B, U, 4), f(B, U), f(B, U)
    half = dev(dh2 * 0.5)
    be.ln_lstm_cell_bwd(half, half, None, dev(dcn), gates, dev(c), cn, chat, iss, dev(gs), dz, dcp, dcnt, B, U)
    close(dcp, dc)
    dzk, dzr = f(B, 4 * U), f(B, 4 * U)
    be.layernorm_bwd(dz, xhk, dev(il(gk, U)), isk, dzk, None, None, B, 4 * U, 4 * U, None)
    be.layernorm_bwd(dz, xhr, dev(il(gr, U)), isr, dzr, None, None, B, 4 * U, 4 * U, None)
    close(unil(dzk.view(B, U, 4).cpu().numpy()) @ W.T, dx)
    close(unil(dzr.view(B, U, 4).cpu().numpy()) @ Ur.T, dh)
    close(unil(dz.cpu().numpy()).sum(0), gl["b"])
    close((dcnt.cpu().numpy().astype(np.float64) * chat.cpu().numpy()).sum(0), gl["gs"], atol=1e-4 * np.abs(gl["gs"]).max())


@pytest.mark.parametrize("B,U,D,masked", [(64, 512, 0, False), (64, 512, 32, False), (5, 16, 3, False),
                                          (20, 32, 0, True)])
def test_lstm_step(be, B, U, D, masked):
    rng = np.random.default_rng(7 + B)
    xz = rng.standard_normal((B, 4 * U)) * 0.5
    h0, c0 = rng.standard_normal((B, U)) * 0.5, rng.standard_normal((B, U)) * 0.5
    Ur = rng.standard_normal((U, 4 * U)) / np.sqrt(U)
    ctx = rng.standard_normal((B, D)) if D else None
    Wc = rng.standard_normal((D, 4 * U)) / np.sqrt(max(D, 1)) if D else None
    T = 3
    ids = rng.integers(0, 3, (B, T)).astype(np.int32) if masked else None
    outp = rng.standard_normal((B, U))
    xz_eff = xz + (ctx @ Wc if D else 0)
    h2, c2, cache = O.lstm_step_fwd(xz_eff, h0, c0, Ur)
    i, f, g, o = cache[:4]
    if masked:
        m = (ids[:, 1] != 0)[:, None]
        hw, cw, ow = np.where(m, h2, h0), np.where(m, c2, c0), np.where(m, h2, outp)
    else:
        hw, cw, ow = h2, c2, h2
    z = lambda: torch.zeros(B, U, device="cuda")
    h, c, out, gates = z(), z(), z(), torch.zeros(B, U, 4, device="cuda")
    be.lstm_step_fwd(dev(il(xz, U)), dev(h0), dev(c0), dev(il(Ur, U)), dev(ctx) if D else None,
                     dev(il(Wc, U)) if D else None, D, dev(ids, torch.int32) if masked else None, T, 1,
                     dev(outp) if masked else None, h, c, out, gates, B, U)
    close(h, hw); close(c, cw); close(out, ow)
    close(gates, np.stack([i, f, g, o], axis=-1))
    # backward: step t given dz_next of step t+1
    dz_next = rng.standard_normal((B, 4 * U)) * 0.3
    da_in, dh_ext, dc_in = rng.standard_normal((B, U)), rng.standard_normal((B, U)), rng.standard_normal((B, U))
    dout_in, dout_t = rng.standard_normal((B, U)), rng.standard_normal((B, U))
    da = da_in + dh_ext + dz_next @ Ur.T
    dout = dout_in + dout_t
    dzo, _, dcp = O.lstm_step_bwd(da + dout, dc_in, cache, Ur)
    if masked:
        dzw = np.where(np.repeat(m, 4 * U, 1), dzo, 0)
        dcw, daw, dow = np.where(m, dcp, dc_in), np.where(m, 0, da), np.where(m, 0, dout)
    else:
        dzw, dcw, daw, dow = dzo, dcp, 0 * da, 0 * dout
    dz, da_o, dc_o, do_o = torch.zeros(B, U, 4, device="cuda"), z(), z(), z()
    be.lstm_step_bwd(dev(il(dz_next, U)), dev(il(Ur, U)), dev(da_in), dev(dh_ext), dev(dc_in), dev(dout_in),
                     dev(dout_t), dev(ids, torch.int32) if masked else None, T, 1, gates, dev(c2), dev(c0), dz, da_o,
                     dc_o, do_o, B, U)
    close(unil(dz.cpu().numpy()), dzw); close(dc_o, dcw); close(da_o, daw); close(do_o, dow)
    # first backward step: no dz_next, optional pointers null
    dz.zero_()
    be.lstm_step_bwd(None, dev(il(Ur, U)), None, None, None, None, dev(dout_t), None, 0, 0, gates, dev(c2), dev(c0),
                     dz, None, dc_o, None, B, U)
    dzo, _, dcp = O.lstm_step_bwd(dout_t, 0 * dc_in, cache, Ur)
    close(unil(dz.cpu().numpy()), dzo); close(dc_o, dcp)


# ---------------------------------------------------------------------- softmax + CCE
def test_softmax_cce(be):
    rng = np.random.default_rng(8)
    rows, V, ld = 96, 5001, 5004
    logits = rng.standard_normal((rows, V)) * 3
    logits[0, 17] = 60.0          # p_y > 1-1e-7 when y = 17 -> clip active
    logits[1, 5] = -60.0          # p_y < 1e-7 when y = 5
    y = rng.integers(0, V, rows).astype(np.int32)
    y[0], y[1] = 17, 5
    logits[2, 100] = logits[2, 200] = logits[2].max() + 1     # tie -> first index
    p = O.softmax(logits)
    ld_ = torch.zeros(rows, ld, device="cuda"); ld_[:, :V] = dev(logits)
    probs, dl = torch.zeros(rows, ld, device="cuda"), torch.zeros(rows, ld, device="cuda")
    loss, corr = torch.zeros(rows, device="cuda"), torch.zeros(rows, device="cuda")
    be.softmax_cce(ld_, dev(y, torch.int32), probs, loss, corr, dl, rows, V, ld, 1.0 / rows)
    close(probs[:, :V], p)
    close(loss, O.cce_from_probs(p, y), rtol=2e-4)
    assert np.array_equal(corr.cpu().numpy(), (p.argmax(-1) == y).astype(np.float32))
    want = O.cce_softmax_bwd(p, y, np.full(rows, 1.0 / rows))
    close(dl[:, :V], want)
    assert (dl[0] == 0).all() and (dl[1] == 0).all()
    # in place
    be.softmax_cce(ld_, dev(y, torch.int32), ld_, loss, corr, None, rows, V, ld, 0.0)
    close(ld_[:, :V], p)
    am = torch.zeros(rows, dtype=torch.int32, device="cuda")
    be.argmax_rows(ld_, am, rows, V, ld)
    assert np.array_equal(am.cpu().numpy(), p.argmax(-1))
    assert am[2].item() == 100


def test_onehot_argmax(be):
    rng = np.random.default_rng(9)
    B, T, V = 4, 3, 50
    ids = rng.integers(0, V, (B, T))
    oh = np.zeros((B, T, V), np.float32)
    np.put_along_axis(oh, ids[..., None], 1.0, -1)
    out = torch.zeros(T * B, dtype=torch.int32, device="cuda")
    be.onehot_argmax(dev(oh), out, B, T, V)
    assert np.array_equal(out.cpu().numpy().reshape(T, B).T, ids)


# -------------------------------------------------------------------------- optimizer
def test_span_sqnorm_lr_reads_only_what_the_update_consumes(be):
    """tnt_span_sqnorm_lr_f32 with the sq_override table: a variable whose clip norm is supplied (>= 0) gets no gradient
    pass (slot 0 of its spans = 0), sum theta^2 is 0 where lambda == 0; without the table every slot is the full pair."""
    from masters_thesis_amd.arena import build_spans
    rng = np.random.default_rng(12)
    lens = [20000, 9000, 4096 * 3 + 5, 64]
    l2 = [0.01, 0.0, 0.0, 3e-5]
    ovr = [-1.0, 4.0, -1.0, 2.0]
    offs, total = [], 0
    for n in lens:
        offs.append(total)
        total += (n + 63) // 64 * 64
    theta, grad = np.zeros(total), np.zeros(total)
    for o, n in zip(offs, lens):
        theta[o:o + n] = rng.standard_normal(n); grad[o:o + n] = rng.standard_normal(n) * 0.01
    sp = build_spans(offs, lens, device="cuda")
    th, gr, l2d = dev(theta), dev(grad), dev(l2)
    at = torch.full((1,), 6, dtype=torch.int64, device="cuda")
    lr, lrt = dev([1e-3]), torch.zeros(1, device="cuda")
    first = sp.first_host
    for table in (None, dev(ovr)):
        part = torch.full((2 * sp.nspan,), 7.0, device="cuda")
        be.span_sqnorm_lr(th, gr, sp.span_seg, sp.span_off, sp.span_len, l2d, part, sp.nspan, at, lr, lrt, 0.9, 0.98, skip=table)
        torch.cuda.synchronize()
        p = part.cpu().numpy().astype(np.float64).reshape(-1, 2)
        for s, (o, n, lam) in enumerate(zip(offs, lens, l2)):
            q, w = p[first[s]:first[s + 1], 0].sum(), p[first[s]:first[s + 1], 1].sum()
            t, g = theta[o:o + n], grad[o:o + n]
            need_g = table is None or ovr[s] < 0
            need_w = table is None or lam != 0
            want_q = ((g + 2 * lam * t) ** 2).sum() if need_g else 0.0
            want_w = (t * t).sum() if need_w else 0.0
            assert abs(q - want_q) <= 1e-5 * max(want_q, 1e-30), (s, q, want_q)
            assert abs(w - want_w) <= 1e-5 * max(want_w, 1e-30), (s, w, want_w)
    want_lr = 1e-3 * np.sqrt(1 - 0.98 ** 7) / (1 - 0.9 ** 7)
    assert abs(float(lrt) - want_lr) <= 1e-6 * want_lr


def test_optimizer(be):
    from masters_thesis_amd.arena import build_spans
    rng = np.random.default_rng(10)
    lens = [20000, 7, 4096 * 3 + 5, 64]
    l2 = [0.01, 0.0, 3e-5, 0.0]
    offs, total = [], 0
    for n in lens:
        offs.append(total)
        total += (n + 63) // 64 * 64
    theta = np.zeros(total); grad = np.zeros(total); m0 = np.zeros(total); v0 = np.zeros(total)
    for o, n in zip(offs, lens):
        theta[o:o + n] = rng.standard_normal(n); grad[o:o + n] = rng.standard_normal(n) * 0.01
        m0[o:o + n] = rng.standard_normal(n) * 0.01; v0[o:o + n] = rng.random(n) * 1e-4
    sp = build_spans(offs, lens, device="cuda")
    nseg = len(lens)
    sq, wsq = torch.zeros(nseg, device="cuda"), torch.zeros(nseg, device="cuda")
    part = torch.zeros(2 * sp.nspan, device="cuda")
    th, gr, md, vd = dev(theta), dev(grad), dev(m0), dev(v0)
    l2d = dev(l2)
    l2o = torch.zeros(1, device="cuda")
    be.seg_sqnorm(th, gr, sp.span_seg, sp.span_off, sp.span_len, sp.seg_first, l2d, part, sq, wsq, l2o, sp.nspan, nseg)
    geff = [grad[o:o + n] + 2 * l * theta[o:o + n] for o, n, l in zip(offs, lens, l2)]
    close(sq, [(g * g).sum() for g in geff], rtol=1e-5)
    close(wsq, [(theta[o:o + n] ** 2).sum() for o, n in zip(offs, lens)], rtol=1e-5)
    close(l2o, [sum(l * (theta[o:o + n] ** 2).sum() for o, n, l in zip(offs, lens, l2))], rtol=1e-5)
    ovr = dev([-1.0, -1.0, -1.0, 25.0])
    state_t = torch.zeros(1, dtype=torch.int64, device="cuda"); state_t += 4
    dstep = torch.zeros(1, dtype=torch.int32, device="cuda")
    lr, lr_t = dev([1e-4]), torch.zeros(1, device="cuda")
    be.step_tick(state_t, dstep, lr, lr_t, 0.9, 0.98)
    assert state_t.item() == 5 and dstep.item() == 1
    close(lr_t, [1e-4 * np.sqrt(1 - 0.98 ** 5) / (1 - 0.9 ** 5)], rtol=1e-6)
    be.adam(th, md, vd, gr, sp.span_seg, sp.span_off, sp.span_len, l2d, sq, ovr, sp.nspan, 0.0, lr_t, 0.9, 0.98, 1e-8,
            0.1)
    for k, (o, n) in enumerate(zip(offs, lens)):
        nrm = np.sqrt((geff[k] ** 2).sum()) if k != 3 else 5.0
        g = geff[k] * 0.1 / max(nrm, 0.1)
        tw, mw, vw = O.adam_update(theta[o:o + n], m0[o:o + n], v0[o:o + n], g, 5)
        close(th[o:o + n], tw, rtol=1e-6); close(md[o:o + n], mw, rtol=1e-5); close(vd[o:o + n], vw, rtol=1e-5)
    # SGD momentum
    th2, mom = dev(theta), dev(m0)
    be.sgd(th2, mom, gr, sp.span_seg, sp.span_off, sp.span_len, l2d, sq, None, sp.nspan, 1e-2, None, 0.9, 0.0)
    for k, (o, n) in enumerate(zip(offs, lens)):
        tw, mw = O.sgd_momentum_update(theta[o:o + n], m0[o:o + n], geff[k], 1e-2)
        close(th2[o:o + n], tw, rtol=1e-6)


@pytest.mark.parametrize("B,K,E,correlated", [(64, 20000, 512, False), (64, 20000, 512, True), (5, 64, 512, False),
                                              (33, 1600, 1024, True)])
def test_dense_gram_norm(be, B, K, E, correlated):
    """Clip-by-norm input of the encoder kernel from the forward's Gram by-products (tnt_dense_fwd_stream_gram_f32 ->
    tnt_dense_gram_norm_f32) against float64 sum (X^T D + 2 l2 W)^2, also for strongly correlated samples with
    gradient rows that sum to zero (what BatchNorm's backward produces), where the Gram form cancels the most."""
    rng = np.random.default_rng(B + K)
    lam, ns = 0.01, 16
    if correlated:
        x = rng.standard_normal((1, K)) * 3 + 0.3 * rng.standard_normal((B, K))
        dpre = rng.standard_normal((B, E)) * 0.01
        dpre -= dpre.mean(0, keepdims=True)
    else:
        x, dpre = rng.standard_normal((B, K)), rng.standard_normal((B, E)) * 0.01
    w = rng.standard_normal((K, E)) / np.sqrt(K)
    bias = 0.1 * rng.standard_normal(E)
    part = torch.zeros(ns * B * E, device="cuda")
    gx, w2 = torch.full((ns * 64 * 64,), 7.0, device="cuda"), torch.full((ns * (E // 32),), 7.0, device="cuda")
    be.dense_fwd_stream_gram(dev(x), dev(w), part, gx, w2, B, E, K, K, E, ns)
    pre = (part.view(ns, B, E).sum(0) + dev(bias)).contiguous()
    close(pre, x @ w + bias, rtol=2e-5)
    close(gx.view(ns, 64, 64).sum(0)[:B, :B], x @ x.T, rtol=1e-5)
    close(w2.sum().reshape(1), [(w * w).sum()], rtol=1e-5)
    nslot = max(4 * B + ns * (E // 32) + 3, 900)
    partial = torch.full((2 * nslot,), 7.0, device="cuda")
    be.dense_gram_norm(dev(dpre), pre, dev(bias), gx, ns, w2, ns * (E // 32), lam, partial, nslot, B, E)
    p = partial.cpu().double().numpy()
    g = x.T @ dpre
    want = ((g + 2 * lam * w) ** 2).sum()
    # the three terms separately: the Gram contraction is the delicate one
    print("norm^2", want, "||g||^2", (g * g).sum(), "gram sum", p[0::2].sum())
    assert abs(p[0::2].sum() - want) <= 2e-5 * want, (p[0::2].sum(), want, (g * g).sum())
    assert abs(p[1::2].sum() - (w * w).sum()) <= 1e-5 * (w * w).sum()


@pytest.mark.parametrize("N,E,Bk,ldx,clip", [(20000, 512, 64, 20000, 0.1), (999, 512, 33, 1000, 0.1), (112, 1024, 7, 112, 0.0)])
def test_dense_dw_fused_norm_and_adam(be, N, E, Bk, ldx, clip):
    """tnt_dense_dw_sqnorm_f32 / tnt_dense_dw_adam_f32 (the encoder kernel's gradient consumed by the optimizer step
    without being written) against the written-out path: dense_dw_skinny -> span norms -> tnt_adam_f32, and float64."""
    from masters_thesis_amd.arena import build_spans
    rng = np.random.default_rng(N + E)
    lam = 0.01
    x = np.zeros((Bk, ldx)); x[:, :N] = rng.standard_normal((Bk, N))
    dpre = rng.standard_normal((Bk, E)) * 0.01
    theta, m0, v0 = rng.standard_normal((N, E)) * 0.05, rng.standard_normal((N, E)) * 1e-3, rng.random((N, E)) * 1e-6
    n = N * E
    sp = build_spans([0], [n], device="cuda")
    l2d = dev([lam])
    # --- written-out path
    g = torch.zeros(N, E, device="cuda")
    be.dense_dw_skinny(dev(x), dev(dpre), g, N, E, Bk, ldx)
    th_a, m_a, v_a = dev(theta), dev(m0), dev(v0)
    sq_a, wsq_a, l2o = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    part = torch.zeros(2 * sp.nspan, device="cuda")
    be.seg_sqnorm(th_a, g, sp.span_seg, sp.span_off, sp.span_len, sp.seg_first, l2d, part, sq_a, wsq_a, l2o, sp.nspan, 1)
    lr_t = dev([3e-4])
    be.adam(th_a, m_a, v_a, g, sp.span_seg, sp.span_off, sp.span_len, l2d, sq_a, None, sp.nspan, 0.0, lr_t, 0.9, 0.98, 1e-8,
            clip)
    # --- fused path
    th_b, m_b, v_b = dev(theta), dev(m0), dev(v0)
    part_b = torch.full((2 * sp.nspan,), 7.0, device="cuda")
    be.dense_dw_sqnorm(dev(x), dev(dpre), th_b, lam, part_b, sp.nspan, N, E, Bk, ldx)
    g64 = x[:, :N].T @ dpre
    q64, w64 = ((g64 + 2 * lam * theta) ** 2).sum(), (theta ** 2).sum()
    pb = part_b.cpu().double().numpy()
    assert abs(pb[0::2].sum() - q64) <= 1e-5 * q64 and abs(pb[1::2].sum() - w64) <= 1e-5 * w64
    close(sq_a, [q64], rtol=1e-5)
    sq_b = dev([pb[0::2].sum()])
    guard = torch.zeros(1, dtype=torch.int32, device="cuda")
    guard[0] = 1                                                                  # tripped guard: nothing moves
    be.dense_dw_adam(dev(x), dev(dpre), th_b, m_b, v_b, lam, sq_b, None, lr_t, 0.9, 0.98, 1e-8, clip, N, E, Bk, ldx, guard=guard)
    assert torch.equal(th_b, dev(theta))
    guard[0] = 0
    be.dense_dw_adam(dev(x), dev(dpre), th_b, m_b, v_b, lam, sq_b, None, lr_t, 0.9, 0.98, 1e-8, clip, N, E, Bk, ldx, guard=guard)
    for a_, b_ in ((th_a, th_b), (m_a, m_b), (v_a, v_b)):
        close(b_, a_.cpu().numpy(), rtol=2e-6)
    cs = clip / max(np.sqrt(q64), clip) if clip > 0 else 1.0
    ge = (g64 + 2 * lam * theta) * cs
    m1 = m0 + (ge - m0) * 0.1
    v1 = v0 + (ge * ge - v0) * 0.02
    close(m_b, m1, rtol=1e-5); close(v_b, v1, rtol=1e-5)
    close(th_b, theta - 3e-4 * m1 / (np.sqrt(v1) + 1e-8), rtol=1e-6)


# ------------------------------------------------------------------- locally dense
@pytest.mark.parametrize("B,N,R,D", [(64, 2000, 36, 32), (3, 37, 4, 16), (64, 3000, 5, 32), (150, 500, 7, 32)])
def test_locally_dense(be, B, N, R, D):
    rng = np.random.default_rng(11)
    groups = tiny_groups(N, R, rng)
    x = rng.standard_normal((B, N))
    Ws = [rng.standard_normal((len(g), D)) / np.sqrt(len(g)) for g in groups]
    bs = [rng.standard_normal(D) * 0.1 for _ in groups]
    goff = np.concatenate([[0], np.cumsum([len(g) for g in groups])]).astype(np.int32)
    idx = np.concatenate(groups).astype(np.int32)
    W = np.concatenate(Ws, axis=0)
    bias = np.stack(bs)
    pre, y = torch.zeros(B, R, D, device="cuda"), torch.zeros(B, R, D, device="cuda")
    xd, idd, gd = dev(x), dev(idx, torch.int32), dev(goff, torch.int32)
    be.locally_dense_fwd(xd, N, idd, gd, dev(W), dev(bias), pre, y, B, R, D, 0.2)
    yo, preo = O.locally_dense_fwd(x, groups, Ws, bs)
    close(pre, preo); close(y, yo)
    dpre = rng.standard_normal((B, R, D))
    dW, db = torch.zeros_like(dev(W)), torch.zeros(R, D, device="cuda")
    be.locally_dense_bwd(xd, N, idd, gd, dev(dpre), dW, db, B, R, D)
    dWo = [x[:, g].T @ dpre[:, r] for r, g in enumerate(groups)]
    close(dW, np.concatenate(dWo, axis=0)); close(db, dpre.sum(0))


# ---------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,R,D,A,U,rate", [(64, 360, 32, 32, 512, 0.2), (3, 4, 5, 3, 16, 0.0), (4, 50, 48, 40, 32, 0.3)])
def test_attention_step(be, B, R, D, A, U, rate):
    rng = np.random.default_rng(12)
    F, h = rng.standard_normal((B, R, D)), rng.standard_normal((B, U)) * 0.5
    W1, b1 = rng.standard_normal((D, A)) / np.sqrt(D), rng.standard_normal(A) * 0.1
    W2, b2 = rng.standard_normal((U, A)) / np.sqrt(U), rng.standard_normal(A) * 0.1
    v, bv = rng.standard_normal((A, 1)), rng.standard_normal(1)
    seed, site_a, site_i, step, lw = 77, 16 + 3, 48 + 3, 5, D + 20
    rate_in = 0.25 if rate > 0 else 0.0
    keep = keep_mask((B, R, A), rate, seed, site_a, step) if rate > 0 else None
    keep_in = keep_mask((B, lw), rate_in, seed, site_i, step)[:, :D] if rate_in > 0 else None
    P, _ = O.attention_proj_fwd(F, W1, b1)
    (ctx, alpha, sd), cache = O.attention_step_fwd(h, F, P, W2, b2, v, bv, keep, rate)
    ctx_d = O.dropout_fwd(ctx, keep_in, rate_in)
    qpre, al, cx, cxd = (torch.zeros(B, A, device="cuda"), torch.zeros(B, R, device="cuda"),
                         torch.zeros(B, D, device="cuda"), torch.zeros(B, D, device="cuda"))
    s_out = torch.zeros(B, R, A, device="cuda")
    Fd, Pd, W2d, vd = dev(F), dev(P), dev(W2), dev(v[:, 0])
    be.attention_step_fwd(dev(h), Fd, Pd, W2d, dev(b2), vd, dev(bv), qpre, al, cx, cxd, s_out, B, R, D, A, U, 0.2, rate,
                          rate_in, lw, seed, site_a, site_i, step)
    close(al, alpha); close(cx, ctx); close(cxd, ctx_d); close(s_out, sd); close(qpre, cache[1])
    dctx_d = rng.standard_normal((B, D))
    dctx = O.dropout_bwd(dctx_d, keep_in, rate_in)
    dh, dF, dsum, dW2, db2, dv, dbv = O.attention_step_bwd(dctx, F, W2, v, cache)
    dP0, dF0 = rng.standard_normal((B, R, A)), rng.standard_normal((B, R, D))
    dPd, dFd, dvb = dev(dP0), dev(dF0), torch.zeros(B, A + 1, device="cuda")
    dq, dhd = torch.zeros(B, A, device="cuda"), torch.zeros(B, U, device="cuda")
    be.attention_step_bwd(dev(dctx_d), Fd, Pd, W2d, vd, qpre, al, dPd, dFd, dvb, dq, dhd, B, R, D, A, U, 0.2, rate,
                          rate_in, lw, seed, site_a, site_i, step)
    close(dhd, dh); close(dPd, dP0 + dsum); close(dFd, dF0 + dF)
    close(dvb[:, :A].sum(0), dv[:, 0], atol=1e-4 * (np.abs(dv).max() + 1))
    close(h.T @ dq.cpu().double().numpy(), dW2, atol=1e-4 * (np.abs(dW2).max() + 1))
    assert abs(dvb[:, A].sum().item()) < 1e-4
    # fresh: the three accumulators are overwritten, whatever they held (the chain's first step replaces a zero fill)
    nan = float("nan")
    dPf, dFf, dvbf = (torch.full((B, R, A), nan, device="cuda"), torch.full((B, R, D), nan, device="cuda"),
                      torch.full((B, A + 1), nan, device="cuda"))
    be.attention_step_bwd(dev(dctx_d), Fd, Pd, W2d, vd, qpre, al, dPf, dFf, dvbf, dq, dhd, B, R, D, A, U, 0.2, rate,
                          rate_in, lw, seed, site_a, site_i, step, fresh=True)
    close(dPf, dsum); close(dFf, dF); assert torch.equal(dvbf, dvb)
    if U % 16 == 0:
        # fused form: dctx_d = dz @ Wc^T computed inside the kernel
        dz = rng.standard_normal((B, 4 * U)) * 0.3
        Wc = rng.standard_normal((D, 4 * U)) / np.sqrt(4 * U)
        dctx2 = O.dropout_bwd(dz @ Wc.T, keep_in, rate_in)
        dh2, dF2, dsum2, _, _, _, _ = O.attention_step_bwd(dctx2, F, W2, v, cache)
        dPd2, dFd2, dvb2 = torch.zeros(B, R, A, device="cuda"), torch.zeros(B, R, D, device="cuda"), torch.zeros(B, A + 1, device="cuda")
        be.attention_step_bwd(None, Fd, Pd, W2d, vd, qpre, al, dPd2, dFd2, dvb2, dq, dhd, B, R, D, A, U, 0.2, rate,
                              rate_in, lw, seed, site_a, site_i, step, None, dev(dz), dev(Wc))
        close(dhd, dh2); close(dPd2, dsum2); close(dFd2, dF2)


def test_dropout_mask4_bit_exact(be):
    """tnt_dropout_mask4_u8: the stored keep bits are the Philox stream of tnt_dropout_f32 / oracle.philox, per site."""
    n, nsites, rate, seed, site0, step = 64 * 90 * 32, 3, 0.2, 1234567890123, 16, 4
    out = torch.zeros(nsites, n // 4, dtype=torch.uint8, device="cuda")
    step_dev = torch.tensor([3], dtype=torch.int32, device="cuda")
    be.dropout_mask4(out, n, nsites, rate, seed, site0, step, step_dev)
    got = out.cpu().numpy()
    for k in range(nsites):
        want = keep_mask((n,), rate, seed, site0 + k, step + 3).reshape(n // 4, 4)
        bits = ((got[k][:, None] >> np.arange(4)[None, :]) & 1).astype(bool)
        assert np.array_equal(bits, want), k
    assert abs(1.0 - np.unpackbits(got).sum() * 2 / (nsites * n) * 0.5 - rate) < 0.01     # ~rate of the bits are 0


@pytest.mark.parametrize("B,R,D,A,U", [(64, 360, 32, 32, 512), (5, 100, 16, 24, 64), (3, 40, 64, 64, 128)])
def test_attention_step_stored_mask_is_bit_identical(be, B, R, D, A, U):
    """The attention step kernels fed with stored keep bits (keep4) against the same kernels running Philox
    themselves: every output bit-identical, forward and backward."""
    rng = np.random.default_rng(14)
    f = lambda *s: torch.tensor(rng.standard_normal(s), dtype=torch.float32, device="cuda")
    h, F, P, W2, b2, v, bv = f(B, U) * 0.5, f(B, R, D), f(B, R, A), f(U, A) * 0.05, f(A) * 0.1, f(A), f(1)
    seed, site_a, site_i, step, lw, rate = 99, 16 + 7, 48 + 7, 2, D + 40, 0.2
    step_dev = torch.tensor([5], dtype=torch.int32, device="cuda")
    keep4 = torch.zeros(1, B * R * A // 4, dtype=torch.uint8, device="cuda")
    be.dropout_mask4(keep4, B * R * A, 1, rate, seed, site_a, step, step_dev)
    outs = []
    for k4 in (None, keep4[0]):
        qpre, al, cx, cxd, s_out = (torch.zeros(B, A, device="cuda"), torch.zeros(B, R, device="cuda"),
                                    torch.zeros(B, D, device="cuda"), torch.zeros(B, D, device="cuda"),
                                    torch.zeros(B, R, A, device="cuda"))
        be.attention_step_fwd(h, F, P, W2, b2, v, bv, qpre, al, cx, cxd, s_out, B, R, D, A, U, 0.2, rate, rate, lw, seed,
                              site_a, site_i, step, step_dev, keep4=k4)
        dP, dF, dvb = torch.ones(B, R, A, device="cuda"), torch.ones(B, R, D, device="cuda"), torch.zeros(B, A + 1, device="cuda")
        dq, dh = torch.zeros(B, A, device="cuda"), torch.zeros(B, U, device="cuda")
        dctx_d = torch.tensor(np.random.default_rng(15).standard_normal((B, D)), dtype=torch.float32, device="cuda")
        be.attention_step_bwd(dctx_d, F, P, W2, v, qpre, al, dP, dF, dvb, dq, dh, B, R, D, A, U, 0.2, rate, rate, lw, seed,
                              site_a, site_i, step, step_dev, keep4=k4)
        outs.append((qpre, al, cx, cxd, s_out, dP, dF, dvb, dq, dh))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert (outs[0][4] == 0).float().mean().item() > 0.1          # the mask really dropped something


@pytest.mark.parametrize("B,masked", [(64, True), (40, True), (64, False), (128, True)])
def test_lstm_seq_fwd_equals_step_kernels(be, B, masked):
    """tnt_lstm_seq_fwd_f32 (one persistent launch for the S dependent steps, XCD-local barriers) against S launches of
    tnt_lstm_step_fwd_f32 driven the way nic.NIC drives them: same arithmetic (states, outputs and gates equal to a few
    float32 ulps -- the two kernels are compiled separately, so fma contraction in the gate math differs), run-to-run
    bit-identical, no barrier timeout."""
    U, T = 512, 15
    S = T + 1
    if not be.lstm_seq_supported(B, U):
        pytest.skip("persistent LSTM kernel not supported on this device (needs 256 CUs, 32 workgroups per XCD)")
    rng = np.random.default_rng(B + masked)
    f = lambda *s: torch.tensor(rng.standard_normal(s), dtype=torch.float32, device="cuda")
    xz, Ur, bl = f(S, B, U, 4) * 0.5, f(U, U, 4) * 0.05, f(U, 4) * 0.1
    cap = rng.integers(1, 50, (B, T)).astype(np.int32)
    for b in range(B):
        cap[b, rng.integers(3, T):] = 0                                  # padding tail: masked steps
    capd = torch.tensor(cap, device="cuda")
    h0, c0 = f(B, U) * 0.3, f(B, U) * 0.3

    def alloc():
        Hs, Cs = torch.zeros(S + 1, B, U, device="cuda"), torch.zeros(S + 1, B, U, device="cuda")
        Hs[0], Cs[0] = h0, c0
        return Hs, Cs, torch.full((T, B, U), 9.0, device="cuda"), torch.zeros(S, B, U, 4, device="cuda")

    Hs, Cs, Out, G = alloc()
    be.lstm_step_fwd(xz[0], Hs[0], Cs[0], Ur, None, None, 0, None, 0, 0, None, Hs[1], Cs[1], None, G[0], B, U, xz_bias=bl)
    for t in range(1, S):
        be.lstm_step_fwd(xz[t], Hs[t], Cs[t], Ur, None, None, 0, capd if masked else None, T, t - 1,
                         Out[t - 2] if (t > 1 and masked) else None, Hs[t + 1], Cs[t + 1], Out[t - 1] if masked else None,
                         G[t], B, U, xz_bias=bl)
    Hs2, Cs2, Out2, G2 = alloc()
    sync = torch.zeros(1025, dtype=torch.int32, device="cuda")
    be.lstm_seq_fwd(xz, Hs2, Cs2, Ur, bl, capd if masked else None, T, 1, Out2 if masked else None, G2, S, B, U, sync)
    torch.cuda.synchronize()
    assert int(sync[1024]) == 0, "a barrier of the persistent kernel timed out"
    for x, y in ((Hs, Hs2), (Cs, Cs2), (G, G2)) + (((Out, Out2),) if masked else ()):
        assert (x - y).abs().max().item() <= 2e-6
    if masked:
        assert torch.equal(Out2 == 9.0, torch.zeros_like(Out2, dtype=torch.bool))     # every output row written
        assert not torch.equal(Hs[S], Hs[S - 3])                        # the sequence really advanced
        keep = torch.tensor(cap[:, -1] == 0, device="cuda")             # rows masked at the last step hold their state
        assert torch.equal(Hs2[S][keep], Hs2[S - 1][keep]) and torch.equal(Out2[T - 1][keep], Out2[T - 2][keep])
    Hs3, Cs3, Out3, G3 = alloc()
    be.lstm_seq_fwd(xz, Hs3, Cs3, Ur, bl, capd if masked else None, T, 1, Out3 if masked else None, G3, S, B, U, sync)
    torch.cuda.synchronize()
    assert torch.equal(Hs2, Hs3) and torch.equal(Cs2, Cs3) and torch.equal(G2, G3)


@pytest.mark.parametrize("B,masked", [(64, True), (40, True), (128, False)])
def test_lstm_seq_fwd_matches_oracle(be, B, masked):
    """tnt_lstm_seq_fwd_f32 straight against the float64 oracle (keras LSTM over the feature step + the masked text
    sequence, NIC.py:138-140): every state, every gate and the masked layer output of all T+1 steps."""
    U, T = 512, 15
    S = T + 1
    if not be.lstm_seq_supported(B, U):
        pytest.skip("persistent LSTM kernel not supported on this device (needs 256 CUs, 32 workgroups per XCD)")
    rng = np.random.default_rng(100 + B)
    xz = rng.standard_normal((S, B, 4 * U)) * 0.5
    Ur = rng.standard_normal((U, 4 * U)) / np.sqrt(U)
    bl = rng.standard_normal(4 * U) * 0.1
    cap = rng.integers(1, 50, (B, T)).astype(np.int32)
    for b in range(B):
        cap[b, rng.integers(2, T):] = 0
    cap[0, 0] = 0                                                       # a row masked from the first text step on
    h, c = rng.standard_normal((B, U)) * 0.3, rng.standard_normal((B, U)) * 0.3
    Hw, Cw, Gw, Ow = [h], [c], [], []
    out = np.zeros((B, U))
    for s in range(S):
        h2, c2, cache = O.lstm_step_fwd(xz[s] + bl, Hw[-1], Cw[-1], Ur)
        Gw.append(np.stack(cache[:4], axis=-1))
        if masked and s >= 1:
            m = (cap[:, s - 1] != 0)[:, None]
            h2, c2 = np.where(m, h2, Hw[-1]), np.where(m, c2, Cw[-1])
            out = np.where(m, h2, out)
            Ow.append(out)
        Hw.append(h2); Cw.append(c2)
    Hs, Cs = torch.zeros(S + 1, B, U, device="cuda"), torch.zeros(S + 1, B, U, device="cuda")
    Hs[0], Cs[0] = dev(h), dev(c)
    Out, G = torch.full((T, B, U), 9.0, device="cuda"), torch.zeros(S, B, U, 4, device="cuda")
    sync = torch.zeros(1025, dtype=torch.int32, device="cuda")
    be.lstm_seq_fwd(dev(il(xz, U)), Hs, Cs, dev(il(Ur, U)), dev(il(bl, U)), dev(cap, torch.int32) if masked else None, T, 1,
                    Out if masked else None, G, S, B, U, sync)
    torch.cuda.synchronize()
    assert int(sync[1024]) == 0, "a barrier of the persistent kernel timed out"
    close(Hs, np.stack(Hw)); close(Cs, np.stack(Cw)); close(G, np.stack(Gw))
    if masked:
        close(Out, np.stack(Ow))


@pytest.mark.parametrize("B,mode", [(64, "nic"), (40, "nic"), (64, "fc"), (128, "plain")])
def test_lstm_seq_bwd_matches_oracle_and_step_kernels(be, B, mode):
    """tnt_lstm_seq_bwd_f32 (the BPTT chain as one persistent launch, weights stationary, partial products pushed through
    the XCD's L2) against (a) the float64 oracle chain of NIC.backward (oracle/models.py, NIC.py:248-249 through the
    keras LSTM of NIC.py:138-140) and (b) the per-step kernel driven the way the models drive it.
    mode nic: feature step + masked text steps (mask_s0 = 1); fc: every step masked (lc_NIC.py:317-318);
    plain: no mask (ThinkAndTell decoder)."""
    U, T = 512, 15
    S = T + 1 if mode != "fc" else T
    s0 = {"nic": 1, "fc": 0, "plain": 0}[mode]
    if not be.lstm_seq_supported(B, U):
        pytest.skip("persistent LSTM kernel not supported on this device")
    rng = np.random.default_rng(200 + B)
    xz = rng.standard_normal((S, B, 4 * U)) * 0.5
    Ur = rng.standard_normal((U, 4 * U)) / np.sqrt(U)
    nseq = S - s0
    cap = rng.integers(1, 50, (B, nseq)).astype(np.int32)
    for b in range(B):
        cap[b, rng.integers(2, nseq):] = 0
    cap[1, 0] = 0
    masked = mode != "plain"
    dOut = rng.standard_normal((nseq, B, U)) * 0.1
    # ---- oracle forward + backward chain
    h, c = np.zeros((B, U)), np.zeros((B, U))
    caches, ms = [], []
    for s in range(S):
        h2, c2, cache = O.lstm_step_fwd(xz[s], h, c, Ur)
        m = (cap[:, s - s0] != 0)[:, None] if (masked and s >= s0) else np.ones((B, 1), bool)
        caches.append(cache); ms.append(m)
        h, c = np.where(m, h2, h), np.where(m, c2, c)
    da, dc, dout = np.zeros((B, U)), np.zeros((B, U)), np.zeros((B, U))
    dzw = np.zeros((S, B, 4 * U))
    for s in reversed(range(S)):
        m = ms[s]
        if s >= s0:
            dout = dout + dOut[s - s0]
        else:
            dout = np.zeros((B, U))
        dz, dh_prev, dc_prev = O.lstm_step_bwd(np.where(m, da + dout, 0), np.where(m, dc, 0), caches[s], Ur)
        dzw[s] = dz
        da = np.where(m, 0, da) + dh_prev
        dc = np.where(m, 0, dc) + dc_prev
        dout = np.where(m, 0, dout)
    # ---- device forward (persistent kernel) to produce gates / cell states
    Hs, Cs = torch.zeros(S + 1, B, U, device="cuda"), torch.zeros(S + 1, B, U, device="cuda")
    Out, G = torch.zeros(nseq, B, U, device="cuda"), torch.zeros(S, B, U, 4, device="cuda")
    sync = torch.zeros(1025, dtype=torch.int32, device="cuda")
    guard = torch.zeros(1, device="cuda")
    capd = dev(cap, torch.int32) if masked else None
    Urd = dev(il(Ur, U))
    be.lstm_seq_fwd(dev(il(xz, U)), Hs, Cs, Urd, None, capd, nseq, s0, Out, G, S, B, U, sync, guard)
    dOutd = dev(dOut)
    dZ = torch.full((S, B, U, 4), 7.0, device="cuda")
    work = torch.zeros(be.lstm_seq_bwd_work_floats(B, U), device="cuda")
    be.lstm_seq_bwd(Urd, dOutd, capd, nseq, s0, G, Cs, dZ, work, S, B, U, sync, guard)
    torch.cuda.synchronize()
    assert int(sync[1024]) == 0 and float(guard) == 0.0
    close(unil(dZ.cpu().numpy()), dzw)
    # ---- the per-step kernel, driven as nic.NIC._bwd_seq_lstm drives it
    dZ2 = torch.zeros(S, B, U, 4, device="cuda")
    z = lambda: torch.zeros(B, U, device="cuda")
    dap, dcp, dop = z(), z(), z()
    for s in range(S - 1, -1, -1):
        first = s == S - 1
        seq = s >= s0
        be.lstm_step_bwd(None if first else dZ2[s + 1], Urd, None if first else dap, None, None if first else dcp,
                         (None if first else dop) if seq else None, dOutd[s - s0] if seq else None,
                         capd if (seq and masked) else None, nseq, s - s0 if seq else 0, G[s], Cs[s + 1], Cs[s], dZ2[s],
                         dap, dcp, dop if seq else None, B, U)
    torch.cuda.synchronize()
    scale = dZ2.abs().max().item()
    assert (dZ - dZ2).abs().max().item() <= 2e-6 * max(1.0, scale) + 1e-6 * scale
    # run-to-run bit-identical (fixed summation order, no atomics)
    dZ3 = torch.zeros_like(dZ)
    be.lstm_seq_bwd(Urd, dOutd, capd, nseq, s0, G, Cs, dZ3, work, S, B, U, sync, guard)
    torch.cuda.synchronize()
    assert torch.equal(dZ, dZ3)


def test_lstm_seq_sync_state_after_graph_replays(be):
    """The persistent kernel re-arms its own sync state (csrc/tnt_seq_sync.h): after 250 replays of a captured launch
    -- no reset node, frozen kernel arguments -- the error word is 0, every ticket / exit counter is back at 0, the
    epoch of every XCD that owns a row block equals the number of launches, all 32 flags of such a group stand at the
    last barrier target of the last launch, and the outputs are bit-identical to the first launch (which also shows that no
    consumer ever took a stale h of the previous launch for this launch's: the sentinel reset protocol of the data-polling
    hand-off)."""
    B, U, T = 64, 512, 15
    S = T + 1
    if not be.lstm_seq_supported(B, U):
        pytest.skip("persistent LSTM kernel not supported on this device")
    rng = np.random.default_rng(5)
    f = lambda *s: torch.tensor(rng.standard_normal(s), dtype=torch.float32, device="cuda")
    xz, Ur, bl = f(S, B, U, 4) * 0.5, f(U, U, 4) * 0.05, f(U, 4) * 0.1
    cap = torch.tensor(rng.integers(0, 3, (B, T)).astype(np.int32), device="cuda")
    Hs, Cs = torch.zeros(S + 1, B, U, device="cuda"), torch.zeros(S + 1, B, U, device="cuda")
    Hs[0], Cs[0] = f(B, U) * 0.3, f(B, U) * 0.3
    Out, G = torch.zeros(T, B, U, device="cuda"), torch.zeros(S, B, U, 4, device="cuda")
    sync = torch.zeros(1025, dtype=torch.int32, device="cuda")
    guard = torch.zeros(1, device="cuda")
    launch = lambda: be.lstm_seq_fwd(xz, Hs, Cs, Ur, bl, cap, T, 1, Out, G, S, B, U, sync, guard)
    launch()
    torch.cuda.synchronize()
    first = (Hs.clone(), Cs.clone(), Out.clone(), G.clone())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch()
    reps = 250
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    st = sync.cpu().numpy().astype(np.int64)
    assert st[1024] == 0 and float(guard) == 0.0
    epochs = st[512 + 2:1024:64]
    active = np.nonzero(epochs)[0]
    # row blocks of 8 rows on all 8 XCDs when the batch fits that way (B <= 64), else of 16 rows
    assert len(active) in ((B + 15) // 16, (B + 7) // 8) and (epochs[active] == 1 + reps).all(), epochs
    assert (st[512:1024:64] == 0).all() and (st[513:1024:64] == 0).all(), "ticket / exit counters not re-armed"
    for x in active:
        flags = st[x * 64:x * 64 + 32]
        # the data-polling forward has ONE flag barrier per launch (behind the reset of hs[1]); the flag-per-step
        # variant (TNT_SEQ_FLAGS=1) has S - 1
        last = (S - 1) if os.environ.get("TNT_SEQ_FLAGS", "0") not in ("", "0") else 1
        assert (flags == (epochs[x] - 1) * 64 + last).all(), (x, flags)
    for a, b in zip(first, (Hs, Cs, Out, G)):
        assert torch.equal(a, b)


def test_lstm_seq_bwd_graph_replays_with_stale_exchange_ring(be):
    """The data-polling hand-off of the persistent BPTT kernel across launches: 200 replays of a captured launch, each
    finding the exchange ring full of the PREVIOUS launch's (valid-looking) tiles and the whole ring overwritten with
    plausible garbage in between -- every result bit-identical to the first launch, error word 0.  A consumer that took a
    stale tile for a fresh one would show up as a different dz."""
    B, U, T = 64, 512, 15
    S = T + 1
    if not be.lstm_seq_supported(B, U):
        pytest.skip("persistent LSTM kernel not supported on this device")
    rng = np.random.default_rng(6)
    f = lambda *s: torch.tensor(rng.standard_normal(s), dtype=torch.float32, device="cuda")
    xz, Ur = f(S, B, U, 4) * 0.5, f(U, U, 4) * 0.05
    cap = torch.tensor(rng.integers(0, 3, (B, T)).astype(np.int32), device="cuda")
    Hs, Cs = torch.zeros(S + 1, B, U, device="cuda"), torch.zeros(S + 1, B, U, device="cuda")
    Out, G = torch.zeros(T, B, U, device="cuda"), torch.zeros(S, B, U, 4, device="cuda")
    sync = torch.zeros(1025, dtype=torch.int32, device="cuda")
    guard = torch.zeros(1, device="cuda")
    be.lstm_seq_fwd(xz, Hs, Cs, Ur, None, cap, T, 1, Out, G, S, B, U, sync, guard)
    dOut = f(T, B, U) * 0.1
    dZ = torch.zeros(S, B, U, 4, device="cuda")
    work = torch.zeros(be.lstm_seq_bwd_work_floats(B, U), device="cuda")
    junk = f(work.numel()) * 0.01
    launch = lambda: be.lstm_seq_bwd(Ur, dOut, cap, T, 1, G, Cs, dZ, work, S, B, U, sync, guard)
    launch()
    torch.cuda.synchronize()
    first = dZ.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch()
    for rep in range(200):
        if rep % 3 == 1:
            work.copy_(junk)          # stale-looking content everywhere, including the buffers reset mid-launch
        dZ.zero_()
        g.replay()
        if rep % 50 == 49:
            torch.cuda.synchronize()
            assert torch.equal(dZ, first), rep
    torch.cuda.synchronize()
    assert torch.equal(dZ, first)
    assert int(sync[1024]) == 0 and float(guard) == 0.0


def test_lstm_seq_guard_codes(be):
    """Device guard of the persistent kernel: a pre-set error word is reported through guard_out and survives; a ticket
    counter that is out of phase (what a launch with a wrong census would leave behind) yields code 2 -- loudly, with
    every wave leaving the kernel -- and after the owner zeroes the state the kernel works again."""
    B, U, T = 64, 512, 3
    S = T + 1
    if not be.lstm_seq_supported(B, U):
        pytest.skip("persistent LSTM kernel not supported on this device")
    rng = np.random.default_rng(6)
    f = lambda *s: torch.tensor(rng.standard_normal(s), dtype=torch.float32, device="cuda")
    xz, Ur = f(S, B, U, 4) * 0.5, f(U, U, 4) * 0.05
    Hs, Cs = torch.zeros(S + 1, B, U, device="cuda"), torch.zeros(S + 1, B, U, device="cuda")
    G = torch.zeros(S, B, U, 4, device="cuda")
    sync = torch.zeros(1025, dtype=torch.int32, device="cuda")
    guard = torch.zeros(1, device="cuda")
    launch = lambda: be.lstm_seq_fwd(xz, Hs, Cs, Ur, None, None, 0, S, None, G, S, B, U, sync, guard)
    launch(); torch.cuda.synchronize()
    good = Hs.clone()
    assert int(sync[1024]) == 0 and float(guard) == 0.0
    sync[1024] = 1                                  # as left behind by an earlier barrier timeout
    launch(); torch.cuda.synchronize()
    assert int(sync[1024]) == 1 and float(guard) == 1.0
    sync.zero_(); guard.zero_()
    sync[512:1024:64] = 7                           # tickets out of phase on every XCD
    launch(); torch.cuda.synchronize()
    assert int(sync[1024]) in (1, 2) and float(guard) != 0.0
    sync.zero_(); guard.zero_()
    Hs[1:].zero_()
    launch(); torch.cuda.synchronize()
    assert int(sync[1024]) == 0 and float(guard) == 0.0 and torch.equal(Hs, good)


@pytest.mark.parametrize("T,B,R,D,A,r_attn,r_in,big", [(4, 20, 100, 32, 32, 0.2, 0.3, 0), (3, 64, 200, 48, 40, 0.0, 0.0, 0),
                                                       (2, 5, 7, 4, 8, 0.25, 0.0, 0), (1, 128, 360, 32, 32, 0.2, 0.2, 0),
                                                       (3, 16, 100, 32, 32, 0.2, 0.3, 1), (3, 16, 100, 32, 32, 0.2, 0.0, 2),
                                                       (3, 16, 100, 32, 32, 0.0, 0.0, 3)])
def test_lc_seq_fwd_equals_step_kernels(be, T, B, R, D, A, r_attn, r_in, big):
    """tnt_lc_seq_fwd_f32 (role-specialised persistent chain: attention workgroups + LSTM workgroups per XCD) against the
    per-step launches it replaces (tnt_attention_step_fwd_f32 + tnt_lstm_step_fwd_f32, themselves oracle-checked above):
    ragged batches (B not a multiple of 16), both attention widths (G4 = 8 / 16), stored keep masks and in-kernel Philox,
    context input dropout.  Same arithmetic up to float32 summation order (gate pre-activations, softmax normalisation).
    big: 1 = |P| beyond 40 (the chain evaluates tanh(P + q) itself instead of through e^{2P} e^{2q}), 2 = |q| beyond 40 in
    some steps (the same, decided per step), 3 = small v (the softmax runs on the pre-known bound, no maximum reduced)."""
    U = 512
    if not be.lstm_seq_supported(B, U):
        pytest.skip("persistent chain kernels not supported on this device")
    rng = np.random.default_rng(T * 1000 + B)
    f = lambda *sh, sc=1.0: dev(rng.standard_normal(sh) * sc)
    F, P, W2, b2, v, bv = (f(B, R, D), f(B, R, A, sc=25.0 if big == 1 else 1.0), f(U, A, sc=(150.0 if big == 2 else 1.0) * U ** -0.5),
                           f(A, sc=0.1), f(A, sc=0.2 if big == 3 else 1.0), f(1))
    xz, Wc, Ur, zb = f(T, B, U, 4, sc=0.5), f(D, U, 4, sc=D ** -0.5), f(U, U, 4, sc=U ** -0.5), f(U, 4, sc=0.1)
    h0, c0 = f(B, U, sc=0.5), f(B, U, sc=0.5)
    seed, s_att, s_in, lw = 4711, 16, 48, D + 20
    step_dev = torch.tensor([3], dtype=torch.int32, device="cuda")
    keep = None
    if r_attn > 0 and T != 2:                        # T == 2 case: masks drawn inside the kernels
        keep = torch.zeros(T, B * R * A // 4, dtype=torch.uint8, device="cuda")
        be.dropout_mask4(keep, B * R * A, T, r_attn, seed, s_att, 0, step_dev)

    def buffers():
        z = lambda *sh: torch.zeros(*sh, device="cuda")
        hs, cs = z(T + 1, B, U), z(T + 1, B, U)
        hs[0], cs[0] = h0, c0
        return dict(hs=hs, cs=cs, gates=z(T, B, U, 4), qpre=z(T, B, A), alpha=z(T, B, R), ctx=z(T, B, D), ctx_d=z(T, B, D))
    ref, got = buffers(), buffers()
    for i in range(T):
        be.attention_step_fwd(ref["hs"][i], F, P, W2, b2, v, bv, ref["qpre"][i], ref["alpha"][i], ref["ctx"][i], ref["ctx_d"][i],
                              None, B, R, D, A, U, 0.2, r_attn, r_in, lw, seed, s_att + i, s_in + i, 0, step_dev,
                              keep4=keep[i] if keep is not None else None)
        be.lstm_step_fwd(xz[i], ref["hs"][i], ref["cs"][i], Ur, ref["ctx_d"][i], Wc, D, None, 0, 0, None, ref["hs"][i + 1],
                         ref["cs"][i + 1], None, ref["gates"][i], B, U, xz_bias=zb)
    sync, guard = torch.zeros(1025, dtype=torch.int32, device="cuda"), torch.zeros(1, device="cuda")
    hd, hd_ref = torch.full((T, B, U), float("nan"), device="cuda"), torch.zeros(T, B, U, device="cuda")
    fwork = torch.full((be.lc_seq_fwd_work_floats(B),), 7.0, device="cuda")        # exchange space: contents irrelevant
    for rep in range(2):                             # the second launch starts from the state the first one left
        # (the second one with the output Dropout riding along: tnt_lc_seq_fwd_drop_f32, one site per step from 77)
        be.lc_seq_fwd(F, P, W2, b2, v, bv, got["qpre"], got["alpha"], got["ctx"], got["ctx_d"], keep,
                      B * R * A // 4 if keep is not None else 0, xz, Wc, Ur, zb, got["hs"], got["cs"], got["gates"], T, B, R, D,
                      A, U, 0.2, r_attn, r_in, lw, seed, s_att, s_in, step_dev, fwork, sync, guard,
                      out_drop=(hd, 0.3, 77) if rep else None)
        torch.cuda.synchronize()
        assert int(sync[1024]) == 0 and float(guard) == 0.0
        if rep:
            be.dropout(got["hs"][1:].view(T * B, U), hd_ref.view(T * B, U), T * B, U, U, 0, U, 0, 0.3, seed, 77, 0, step_dev,
                       rows_per_site=B)
            assert torch.equal(hd, hd_ref) and 0.2 < float((hd == 0).float().mean()) < 0.4
        # (big == 2: |q| runs to a few hundred, one ulp of q is ~2e-5 there, and the two paths add its 512 products in different
        # orders -- 16 MFMA partials against one serial sum -- so the query itself differs by that much before tanh sees it)
        qs = max(1.0, ref["qpre"].abs().max().item() / 4.0) if big == 2 else 1.0
        for k in ref:
            d = (got[k] - ref[k]).abs().max().item()
            assert d <= 2e-5 * qs * max(1.0, ref[k].abs().max().item()), (k, d)


@pytest.mark.parametrize("T,B,R,D,A,r_attn,r_in,mse,big", [(4, 20, 100, 32, 32, 0.2, 0.3, 0.0, 0), (3, 64, 200, 48, 40, 0.0, 0.0, 0.01, 0),
                                                           (2, 5, 7, 4, 8, 0.25, 0.0, 0.0, 0), (1, 128, 360, 32, 32, 0.2, 0.2, 0.0, 0),
                                                           (15, 64, 360, 32, 32, 0.2, 0.2, 0.0, 0), (3, 16, 100, 32, 32, 0.2, 0.3, 0.0, 1),
                                                           (3, 16, 100, 32, 32, 0.2, 0.0, 0.0, 2)])
def test_lc_seq_bwd_equals_step_kernels(be, T, B, R, D, A, r_attn, r_in, mse, big):
    """tnt_lc_seq_bwd_f32 (the backward chain as one persistent launch: LSTM-backward workgroups pushing partial da tiles,
    attention-backward workgroups with dP / dF accumulated in registers) against the per-step launches it replaces
    (tnt_lstm_step_bwd_f32 with context-gradient parts + tnt_attention_step_bwd_f32, themselves oracle-checked above):
    dz of every step, dqpre, and the accumulated dP / dF / dvb.  Ragged batches, both attention widths, stored and in-kernel
    masks, context input dropout, the attention-MSE term.  Two launches: the second starts from the ring state of the first.
    big: 1 = |P| beyond 40, 2 = |q| beyond 40 in some steps (the chain then recomputes tanh(P + q) itself, not through
    e^{2P} e^{2q})."""
    U = 512
    if not be.lstm_seq_supported(B, U):
        pytest.skip("persistent chain kernels not supported on this device")
    rng = np.random.default_rng(T * 1000 + B + 1)
    f = lambda *sh, sc=1.0: dev(rng.standard_normal(sh) * sc)
    F, P, W2, v = f(B, R, D), f(B, R, A, sc=25.0 if big == 1 else 1.0), f(U, A, sc=U ** -0.5), f(A)
    Wc, Ur = f(D, U, 4, sc=D ** -0.5), f(U, U, 4, sc=U ** -0.5)
    alpha = dev(O.softmax(rng.standard_normal((T, B, R)), axis=-1))
    qpre, dout = f(T, B, A, sc=30.0 if big == 2 else 1.0), f(T, B, U, sc=0.1)
    gates = torch.sigmoid(f(T, B, U, 4))
    gates[..., 2] = gates[..., 2] * 2 - 1                    # the candidate gate is a tanh
    cs = f(T + 1, B, U, sc=0.5)
    seed, s_att, s_in, lw = 4711, 16, 48, D + 20
    step_dev = torch.tensor([3], dtype=torch.int32, device="cuda")
    dout_raw = dout
    dout = torch.empty_like(dout_raw)                        # Dropout' behind the LSTM: one site per step from 77
    be.dropout(dout_raw.view(T * B, U), dout.view(T * B, U), T * B, U, U, 0, U, 0, 0.3, seed, 77, 0, step_dev, rows_per_site=B)
    keep = None
    if r_attn > 0 and T != 2:
        keep = torch.zeros(T, B * R * A // 4, dtype=torch.uint8, device="cuda")
        be.dropout_mask4(keep, B * R * A, T, r_attn, seed, s_att, 0, step_dev)
    z = lambda *sh: torch.zeros(*sh, device="cuda")
    # ---- per-step reference
    r_dz, r_dq, r_dP, r_dF, r_dvb = z(T, B, U, 4), z(T, B, A), z(B, R, A), z(B, R, D), z(B, A + 1)
    dh_att, dc, parts = z(B, U), z(B, U), z(U // 16, B, D)
    for i in range(T - 1, -1, -1):
        last = i == T - 1
        use_parts = (U // 16) * D <= 1024                  # the per-step kernels' limit; else dctx = dz Wc^T inside the attention step
        be.lstm_step_bwd(None if last else r_dz[i + 1], Ur, None, None if last else dh_att, None if last else dc, None,
                         dout[i], None, 0, 0, gates[i], cs[i + 1], cs[i], r_dz[i], None, dc, None, B, U,
                         Wc=Wc if use_parts else None, D=D, dctx_part=parts if use_parts else None)
        kw = dict(dctx_part=parts, nparts=U // 16) if use_parts else dict(dz=r_dz[i], Wc=Wc)
        be.attention_step_bwd(None, F, P, W2, v, qpre[i], alpha[i], r_dP, r_dF, r_dvb, r_dq[i], dh_att, B, R, D, A, U, 0.2,
                              r_attn, r_in, lw, seed, s_att + i, s_in + i, 0, step_dev,
                              keep4=keep[i] if keep is not None else None, alpha_mse=mse, fresh=last, **kw)
    # ---- one launch
    work = torch.full((be.lc_seq_bwd_work_floats(B, U),), 7.0, device="cuda")
    sync, guard = torch.zeros(1025, dtype=torch.int32, device="cuda"), torch.zeros(1, device="cuda")
    nan = float("nan")
    for rep in range(2):
        g_dz, g_dq = torch.full((T, B, U, 4), nan, device="cuda"), torch.full((T, B, A), nan, device="cuda")
        g_dP, g_dF, g_dvb = (torch.full((B, R, A), nan, device="cuda"), torch.full((B, R, D), nan, device="cuda"),
                             torch.full((B, A + 1), nan, device="cuda"))
        # (the second launch receives the UNMASKED output gradient and applies Dropout' itself: tnt_lc_seq_bwd_drop_f32)
        be.lc_seq_bwd(F, P, W2, v, qpre, alpha, keep, B * R * A // 4 if keep is not None else 0, g_dP, g_dF, g_dvb, g_dq, Ur, Wc,
                      dout_raw if rep else dout, gates, cs, g_dz, work, T, B, R, D, A, U, 0.2, r_attn, r_in, lw, seed, s_att,
                      s_in, step_dev, mse, sync, guard, out_drop=(0.3, 77) if rep else None)
        torch.cuda.synchronize()
        assert int(sync[1024]) == 0 and float(guard) == 0.0
        for name, got, ref in (("dz", g_dz, r_dz), ("dqpre", g_dq, r_dq), ("dP", g_dP, r_dP), ("dF", g_dF, r_dF),
                               ("dvb", g_dvb[:, :A], r_dvb[:, :A])):
            d = (got - ref).abs().max().item()
            assert d <= 3e-5 * max(1e-3, ref.abs().max().item()), (rep, name, d, ref.abs().max().item())
        assert g_dvb[:, A].abs().max().item() < 1e-4


@pytest.mark.parametrize("T,B,R", [(5, 8, 30), (15, 64, 360), (3, 5, 129)])
def test_attention_metric(be, T, B, R):
    rng = np.random.default_rng(13)
    alpha = O.softmax(rng.standard_normal((T, B, R)), axis=-1)
    want = ((1 - alpha.sum(1)) ** 2).mean()
    out = torch.zeros(1, device="cuda")
    npart = be.attention_metric_parts(T, R)
    be.attention_metric(dev(alpha), out, torch.zeros(npart, device="cuda"), T, B, R)
    close(out, [want])
    # partials only (out = None), totalled by the step-finalize launch's third job
    work, out2 = torch.full((npart + 3,), 7.0, device="cuda"), torch.zeros(1, device="cuda")
    be.attention_metric(dev(alpha), None, work, T, B, R)
    assert (work[npart:] == 7.0).all()
    z = torch.zeros(1, device="cuda")
    be.step_finalize(z, torch.zeros(1, dtype=torch.int32, device="cuda"), z, z, z, None, 0, x2=work, out2=out2, n2=npart,
                     scale2=1.0 / (T * R))
    close(out2, [want])


@pytest.mark.parametrize("V,ld,from_logits,temp", [(11, 12, True, 1.0), (5001, 5004, False, 1.0), (5001, 5004, True, 0.7),
                                                   (300, 300, False, 2.0)])
def test_sample_rows(be, V, ld, from_logits, temp):
    """Categorical sampling: ids equal the oracle's inverse-CDF pick except where u*sum sits within float32
    rounding of a CDF edge; the empirical distribution follows the probabilities."""
    rng = np.random.default_rng(91)
    rows = 512
    logits = rng.standard_normal((rows, ld)) * 2.0
    p = np.exp(logits[:, :V]); p /= p.sum(-1, keepdims=True)
    p[:, 3] = 0.0                                            # zero-probability class is never drawn
    x = logits.copy() if from_logits else np.zeros((rows, ld))
    if not from_logits:
        x[:, :V] = p
    out = torch.zeros(rows, dtype=torch.int32, device="cuda")
    step_dev = torch.tensor([2], dtype=torch.int32, device="cuda")
    be.sample_rows(dev(x), out, rows, V, ld, temp, from_logits, 1234, 7, 3, step_dev)
    got = out.cpu().numpy()
    want, margin = O.sample_rows(x[:, :V], temp, from_logits, 1234, 7, 5)
    bad = got != want
    assert np.all(margin[bad] < 1e-5), (int(bad.sum()), margin[bad].max() if bad.any() else 0)
    assert bad.mean() < 0.01
    assert (got >= 0).all() and (got < V).all()
    if not from_logits:
        assert not (got == 3).any()
    # same (seed, site, step) -> same draw; another step -> another draw
    out2 = torch.zeros_like(out)
    be.sample_rows(dev(x), out2, rows, V, ld, temp, from_logits, 1234, 7, 5, None)
    assert torch.equal(out, out2)
    be.sample_rows(dev(x), out2, rows, V, ld, temp, from_logits, 1234, 7, 6, None)
    assert not torch.equal(out, out2)


def test_sample_rows_distribution(be):
    V, rows = 6, 60000
    p = np.array([0.05, 0.3, 0.0, 0.4, 0.2, 0.05])
    x = np.tile(p, (rows, 1))
    out = torch.zeros(rows, dtype=torch.int32, device="cuda")
    be.sample_rows(dev(x), out, rows, V, V, 1.0, False, 99, 1, 0, None)
    freq = np.bincount(out.cpu().numpy(), minlength=V) / rows
    assert np.abs(freq - p).max() < 0.01


@pytest.mark.parametrize("rows,C,training,rates", [(64, 512, True, (0.2, 0.3)), (3, 36, True, (0.0, 0.0)),
                                                   (200, 96, True, (0.1, 0.0)), (64, 512, False, (0.2, 0.3))])
def test_enc_tail_fused_equals_composition(be, rows, C, training, rates):
    """tnt_enc_tail_{fwd,bwd}: the one-launch encoder tail against the generic entry points it replaces
    (dropout -> batchnorm -> dropout; and the reverse chain + LeakyReLU' + bias gradient) and the oracle."""
    rng = np.random.default_rng(95)
    r_feat, r_lstm = rates
    y = rng.standard_normal((rows, C)) * 2 + 0.5
    gamma, beta = 1 + 0.1 * rng.standard_normal(C), 0.1 * rng.standard_normal(C)
    mm0, mv0 = 0.1 * rng.standard_normal(C), 1 + 0.1 * rng.random(C)
    step_dev = torch.tensor([4], dtype=torch.int32, device="cuda")
    seed, sf, sl = 77, 2, 48
    f = lambda *s: torch.zeros(*s, dtype=torch.float32, device="cuda")
    work = f(C * (2 * be.bn_nchunk(rows) + 1))
    # --- composition
    mm_a, mv_a = dev(mm0), dev(mv0)
    yd = dev(y).clone()
    if training and r_feat > 0:
        be.dropout(yd, yd, rows, C, C, 0, C, 0, r_feat, seed, sf, 0, step_dev)
    out_a, xhat_a, inv_a = f(rows, C), f(rows, C), f(max(rows, C))
    be.batchnorm_fwd(yd, dev(gamma), dev(beta), mm_a, mv_a, out_a, xhat_a, inv_a, rows, C, C, training, 1e-3, 0.99, work)
    if training and r_lstm > 0:
        be.dropout(out_a, out_a, rows, C, C, 0, C, 0, r_lstm, seed, sl, 0, step_dev)
    # --- fused
    mm_b, mv_b = dev(mm0), dev(mv0)
    out_b, xhat_b, inv_b = f(rows, C), f(rows, C), f(max(rows, C))
    be.enc_tail_fwd(dev(y), dev(gamma), dev(beta), mm_b, mv_b, out_b, xhat_b, inv_b, rows, C, C, training, 1e-3, 0.99,
                    r_feat if training else 0.0, r_lstm if training else 0.0, seed, sf, sl, step_dev)
    for a_, b_ in ((out_a, out_b), (xhat_a, xhat_b), (inv_a[:C], inv_b[:C]), (mm_a, mm_b), (mv_a, mv_b)):
        close(b_, a_.cpu().numpy(), rtol=2e-5)
    assert torch.equal(out_a == 0, out_b == 0)                       # identical dropout pattern
    if not training:
        return
    # oracle (float64) for the forward statistics
    want, cache, mm_w, mv_w = O.batchnorm_fwd(yd.cpu().double().numpy(), gamma, beta, mm0, mv0, True)
    close(xhat_b, cache[0])
    close(mm_b, mm_w); close(mv_b, mv_w)
    # --- backward
    dout = rng.standard_normal((rows, C))
    pre = rng.standard_normal((rows, C))
    d = dev(dout).clone()
    if r_lstm > 0:
        be.dropout(d, d, rows, C, C, 0, C, 0, r_lstm, seed, sl, 0, step_dev)
    dx, dg_a, db_a = f(rows, C), f(C), f(C)
    be.batchnorm_bwd(d, xhat_a, dev(gamma), inv_a, dx, dg_a, db_a, rows, C, C, True, work)
    if r_feat > 0:
        be.dropout(dx, dx, rows, C, C, 0, C, 0, r_feat, seed, sf, 0, step_dev)
    dpre_a, dbias_a = f(rows, C), f(C)
    be.act_bwd(dev(pre), dx, dpre_a, rows * C, 1, 0.2)
    be.colsum(dpre_a, dbias_a, rows, C, C, work)
    dpre_b, dg_b, db_b, dbias_b = f(rows, C), f(C), f(C), f(C)
    be.enc_tail_bwd(dev(dout), xhat_b, dev(gamma), inv_b, dev(pre), dpre_b, dg_b, db_b, dbias_b, rows, C, C, r_feat,
                    r_lstm, 0.2, seed, sf, sl, step_dev)
    for a_, b_ in ((dpre_a, dpre_b), (dg_a, dg_b), (db_a, db_b), (dbias_a, dbias_b)):
        close(b_, a_.cpu().numpy(), rtol=5e-5)


def test_embedding_fwd_drop_stage_sum2(be):
    rng = np.random.default_rng(96)
    B, T, E, V, N, U = 5, 7, 24, 31, 40, 16
    table = rng.standard_normal((V, E))
    ids = rng.integers(0, V, (B, T)).astype(np.int32)
    step_dev = torch.tensor([3], dtype=torch.int32, device="cuda")
    plain, dropped = torch.zeros(T * B, E, device="cuda"), torch.zeros(T * B, E, device="cuda")
    be.embedding_fwd_drop(dev(table), dev(ids, torch.int32), plain, dropped, B, T, E, E, V, 0.3, 11, 49, 0, step_dev)
    want = O.embedding_fwd(table, ids)                                    # (B,T,E)
    close(plain.view(T, B, E).permute(1, 0, 2), want)
    k = keep_mask((B, T, E), 0.3, 11, 49, 3)
    close(dropped.view(T, B, E).permute(1, 0, 2), np.where(k, want / np.float32(0.7), 0.0), rtol=1e-6)
    ref = torch.zeros(T * B, E, device="cuda")
    be.dropout(plain, ref, T * B, E, E, B, E, 0, 0.3, 11, 49, 0, step_dev)  # the two-launch composition
    assert torch.equal(ref, dropped)
    # --- staging of a device-resident batch
    x = dev(rng.standard_normal((B, N))); cap = dev(ids, torch.int32); tgt = dev(rng.integers(0, V, (B, T)), torch.int32)
    a0, c0 = dev(rng.standard_normal((B, U))), dev(rng.standard_normal((B, U)))
    xd, capd, tgtd = torch.zeros(B, N, device="cuda"), torch.zeros(B, T, dtype=torch.int32, device="cuda"), \
        torch.zeros(T * B, dtype=torch.int32, device="cuda")
    h0, c0d = torch.zeros(B, U, device="cuda"), torch.zeros(B, U, device="cuda")
    xT = torch.full((N, 8), 7.0, device="cuda")
    be.stage_batch(x, xd, cap, capd, tgt, tgtd, a0, h0, c0, c0d, B, T, N, N, U, xT, 8)
    assert torch.equal(xd, x) and torch.equal(capd, cap) and torch.equal(h0, a0) and torch.equal(c0d, c0)
    assert torch.equal(xT[:, :B], x.t()) and float(xT[:, B:].abs().sum()) == 0.0
    assert torch.equal(tgtd.view(T, B), tgt.t())
    # ... with tnt_dropout_mask4_u8's job riding in the same launch (tnt_stage_batch_masks_f32): same copies, same mask bytes
    nm, ns = 4 * 1000, 3
    mk_ref, mk = torch.zeros(ns, nm // 4, dtype=torch.uint8, device="cuda"), torch.full((ns, nm // 4), 255, dtype=torch.uint8, device="cuda")
    be.dropout_mask4(mk_ref, nm, ns, 0.2, 99, 16, 0, step_dev)
    xd.zero_(); xT.fill_(7.0); capd.zero_(); h0.zero_()
    be.stage_batch(x, xd, cap, capd, tgt, tgtd, a0, h0, c0, c0d, B, T, N, N, U, xT, 8, masks=(mk, nm, ns, 0.2, 99, 16, step_dev))
    assert torch.equal(mk, mk_ref)
    assert torch.equal(xd, x) and torch.equal(capd, cap) and torch.equal(h0, a0) and torch.equal(xT[:, :B], x.t())
    x2 = dev(rng.standard_normal((B, 37))); xd2 = torch.zeros(B, 40, device="cuda")      # ragged width, padded rows
    be.stage_batch(x2, xd2, cap, capd, None, tgtd, a0, h0, c0, c0d, B, T, 37, 40, U)
    assert torch.equal(xd2[:, :37], x2) and float(xd2[:, 37:].abs().sum()) == 0.0
    # --- the same staging from IEEE-half betas ("fp16 on-wire"): exactly the float widening of the half values
    for (xs, width, ld, xTt) in ((x, N, N, True), (x2, 37, 40, False)):
        xh = xs.half()
        xd3 = torch.zeros(B, ld, device="cuda")
        xT3 = torch.full((width, 8), 7.0, device="cuda") if xTt else None
        be.stage_batch(xh, xd3, cap, capd, tgt, tgtd, a0, h0, c0, c0d, B, T, width, ld, U, xT3, 8 if xTt else 0)
        assert torch.equal(xd3[:, :width], xh.float()) and float(xd3[:, width:].abs().sum()) == 0.0
        if xTt:
            assert torch.equal(xT3[:, :B], xh.float().t()) and float(xT3[:, B:].abs().sum()) == 0.0
        assert torch.equal(capd, cap) and torch.equal(h0, a0) and torch.equal(tgtd.view(T, B), tgt.t())
    # --- two sums in one launch
    v0, v1 = dev(rng.standard_normal(960)), dev(rng.standard_normal(960))
    out = torch.zeros(2, device="cuda")
    be.sum2(v0, out[0:1], v1, out[1:2], 960, 0.5)
    close(out, [0.5 * v0.double().sum().item(), 0.5 * v1.double().sum().item()], rtol=1e-5)


@pytest.mark.parametrize("N,E,Bk,ldx", [(20000, 512, 64, 20000), (37, 32, 3, 40), (1000, 96, 17, 1000), (2000, 64, 8, 2000),
                                         (1001, 256, 33, 1004), (999, 512, 64, 1000)])
def test_dense_dw_skinny(be, N, E, Bk, ldx):
    """dW = X^T dpre of the dense encoder (persistent skinny-K kernel) against float64 and the generic GEMM."""
    rng = np.random.default_rng(97)
    x = np.zeros((Bk, ldx)); x[:, :N] = rng.standard_normal((Bk, N))
    dpre = rng.standard_normal((Bk, E)) * 0.01
    dw = torch.full((N, E), 7.0, device="cuda")
    be.dense_dw_skinny(dev(x), dev(dpre), dw, N, E, Bk, ldx)
    close(dw, x[:, :N].T @ dpre)
    ref = torch.zeros(N, E, device="cuda")
    be.gemm(dev(x), dev(dpre), ref, N, E, Bk, ldx, E, E, transA=True)
    close(dw, ref.cpu().numpy(), rtol=2e-6)


@pytest.mark.parametrize("B,K,E,ldx,ns", [(64, 20000, 512, 20000, 16), (5, 100, 32, 104, 3), (130, 1028, 64, 1028, 16),
                                          (64, 36, 96, 36, 16), (33, 4, 32, 4, 1)])
@pytest.mark.parametrize("training", [True, False])
def test_dense_fwd_stream_and_tail(be, B, K, E, ldx, ns, training):
    """Streaming skinny-M encoder forward (K-split partials) + the tail that sums them, against float64 / the oracle and
    against the path they replace (generic GEMM with the bias + LeakyReLU epilogue -> tnt_enc_tail_fwd_f32)."""
    rng = np.random.default_rng(98)
    x = np.zeros((B, ldx)); x[:, :K] = rng.standard_normal((B, K))
    w = rng.standard_normal((K, E)) / np.sqrt(K)
    bias = 0.1 * rng.standard_normal(E)
    part = torch.full((ns * B * E,), 7.0, device="cuda")
    be.dense_fwd_stream(dev(x), dev(w), part, B, E, K, ldx, E, ns)
    pre64 = x[:, :K] @ w + bias
    close(part.view(ns, B, E).sum(0) + dev(bias), pre64, rtol=2e-5)       # f32 sums over K = 20000
    gamma, beta = 1 + 0.1 * rng.standard_normal(E), 0.1 * rng.standard_normal(E)
    mm0, mv0 = 0.1 * rng.standard_normal(E), 1 + 0.1 * rng.random(E)
    step_dev = torch.tensor([4], dtype=torch.int32, device="cuda")
    f = lambda *s: torch.zeros(*s, dtype=torch.float32, device="cuda")
    rates = (0.1, 0.2) if training else (0.0, 0.0)
    # --- the path it replaces
    pre_a, y_a = f(B, E), f(B, E)
    be.gemm(dev(x), dev(w), y_a, B, E, K, ldx, E, E, bias=dev(bias), pre=pre_a, act=1, slope=0.2)
    mm_a, mv_a, out_a, xhat_a, inv_a = dev(mm0), dev(mv0), f(B, E), f(B, E), f(max(B, E))
    be.enc_tail_fwd(y_a, dev(gamma), dev(beta), mm_a, mv_a, out_a, xhat_a, inv_a, B, E, E, training, 1e-3, 0.99, *rates,
                    77, 2, 48, step_dev)
    # --- partial-summing tail
    pre_b = f(B, E)
    mm_b, mv_b, out_b, xhat_b, inv_b = dev(mm0), dev(mv0), f(B, E), f(B, E), f(max(B, E))
    be.enc_tail_fwd_sk(part, ns, dev(bias), pre_b, 0.2, dev(gamma), dev(beta), mm_b, mv_b, out_b, xhat_b, inv_b, B, E, E,
                       training, 1e-3, 0.99, *rates, 77, 2, 48, step_dev)
    close(pre_b, pre64, rtol=2e-5)
    close(pre_b, pre_a.cpu().numpy(), rtol=2e-5)
    # BatchNorm divides by the batch deviation, which amplifies the summation-order difference of the two products
    for a_, b_ in ((out_a, out_b), (xhat_a, xhat_b), (inv_a[:E], inv_b[:E]), (mm_a, mm_b), (mv_a, mv_b)):
        close(b_, a_.cpu().numpy(), rtol=2e-4, atol=2e-4)
    assert torch.equal(out_a == 0, out_b == 0)                       # identical dropout pattern
    y64 = np.where(pre64 > 0, pre64, 0.2 * pre64)
    if not training:
        close(out_b, (y64 - mm0) / np.sqrt(mv0 + 1e-3) * gamma + beta, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("B,N,R,D,piece", [(64, 2000, 36, 32, 64), (5, 300, 7, 16, 8), (150, 900, 9, 32, 50)])
def test_locally_dense_split_equals_unsplit(be, B, N, R, D, piece):
    """Split mode (pieces of <= `piece` voxels per workgroup) of the region-wise encoder against the one-workgroup-
    per-region entry points and the oracle."""
    rng = np.random.default_rng(12)
    groups = tiny_groups(N, R, rng)
    goff = np.concatenate([[0], np.cumsum([len(g) for g in groups])]).astype(np.int32)
    vg, vr, vf, rf = [0], [], [], [0]
    for r in range(R):
        k = int(goff[r])
        while True:
            k2 = min(int(goff[r + 1]), k + piece)
            vg.append(k2); vr.append(r); vf.append(int(k == goff[r])); k = k2
            if k >= goff[r + 1]:
                break
        rf.append(len(vr))
    NV = len(vr)
    assert NV > R
    x = rng.standard_normal((B, N))
    Ws = [rng.standard_normal((len(g), D)) / np.sqrt(len(g)) for g in groups]
    bs = [0.1 * rng.standard_normal(D) for _ in groups]
    ti = lambda a: dev(np.asarray(a), torch.int32)
    idx, Wc, bc = ti(np.concatenate(groups)), dev(np.concatenate(Ws)), dev(np.stack(bs))
    f = lambda *s: torch.zeros(*s, dtype=torch.float32, device="cuda")
    pre, y, part = f(B, R, D), f(B, R, D), f(NV, 64, D)
    be.locally_dense_fwd_split(dev(x), N, idx, ti(vg), ti(vr), ti(rf), NV, Wc, bc, pre, y, part, B, R, D, 0.2)
    want_y, want_pre = O.locally_dense_fwd(x, groups, Ws, bs)
    close(pre, want_pre); close(y, want_y)
    pre2, y2 = f(B, R, D), f(B, R, D)
    be.locally_dense_fwd(dev(x), N, idx, ti(goff), Wc, bc, pre2, y2, B, R, D, 0.2)
    close(pre, pre2.cpu().numpy(), rtol=1e-5)
    dpre = rng.standard_normal((B, R, D))
    dW, db, dW2, db2 = f(int(goff[-1]), D), f(R, D), f(int(goff[-1]), D), f(R, D)
    be.locally_dense_bwd_split(dev(x), N, idx, ti(vg), ti(vr), ti(vf), NV, dev(dpre), dW, db, B, R, D)
    be.locally_dense_bwd(dev(x), N, idx, ti(goff), dev(dpre), dW2, db2, B, R, D)
    close(dW, dW2.cpu().numpy(), rtol=1e-5); close(db, db2.cpu().numpy(), rtol=1e-5)
    close(db, dpre.sum(0))
    # voxel-major betas (xT[N][ldt]): same results, coalesced gather
    ldt = (B + 3) // 4 * 4
    xT = torch.zeros(N, ldt, device="cuda"); xT[:, :B] = dev(x).t()
    pre3, y3, dW3, db3 = f(B, R, D), f(B, R, D), f(int(goff[-1]), D), f(R, D)
    be.locally_dense_fwd_split(xT, ldt, idx, ti(vg), ti(vr), ti(rf), NV, Wc, bc, pre3, y3, part, B, R, D, 0.2, voxel_major=True)
    be.locally_dense_bwd_split(xT, ldt, idx, ti(vg), ti(vr), ti(vf), NV, dev(dpre), dW3, db3, B, R, D, voxel_major=True)
    assert torch.equal(pre3, pre) and torch.equal(y3, y) and torch.equal(dW3, dW) and torch.equal(db3, db)


@pytest.mark.parametrize("B,U,D,with_next", [(64, 512, 32, True), (64, 512, 32, False), (5, 48, 16, True), (20, 256, 64, True)])
def test_lstm_bwd_context_partials(be, B, U, D, with_next):
    """tnt_lstm_step_bwd_f32 leaves dctx partials per 16-unit block; their sum is dz @ Wc^T (what the attention
    backward otherwise computes itself), and dz is unchanged by asking for them."""
    rng = np.random.default_rng(98)
    f = lambda *s: torch.zeros(*s, dtype=torch.float32, device="cuda")
    Ur = dev(il(rng.standard_normal((U, 4 * U)) / np.sqrt(U), U))
    Wc = dev(il(rng.standard_normal((D, 4 * U)) / np.sqrt(D), U))
    gates = dev(rng.uniform(0.1, 0.9, (B, U, 4)))
    c, cprev = dev(rng.standard_normal((B, U))), dev(rng.standard_normal((B, U)))
    dz_next = dev(il(rng.standard_normal((B, 4 * U)) * 0.1, U)) if with_next else None
    dh_ext, dc_in = dev(rng.standard_normal((B, U))), dev(rng.standard_normal((B, U)))
    outs = []
    for want_parts in (False, True):
        dz, dc_out = f(B, U, 4), f(B, U)
        parts = f(U // 16, B, D) if want_parts else None
        be.lstm_step_bwd(dz_next, Ur, None, dh_ext, dc_in, None, None, None, 0, 0, gates, c, cprev, dz, None, dc_out, None,
                         B, U, Wc=Wc if want_parts else None, D=D, dctx_part=parts)
        outs.append((dz, dc_out, parts))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    dz, parts = outs[1][0], outs[1][2]
    want = dz.view(B, 4 * U).double().cpu().numpy() @ Wc.view(D, 4 * U).double().cpu().numpy().T
    close(parts.sum(0), want)


@pytest.mark.parametrize("M,N,K,tA,tB,pad", [
    (960, 512, 5001, 0, 1, 3), (512, 5001, 960, 1, 0, 3), (1024, 2048, 512, 0, 0, 0), (512, 2048, 1024, 1, 0, 0),
    (33, 17, 29, 0, 0, 0), (33, 17, 29, 0, 1, 2), (33, 17, 29, 1, 0, 1), (5, 7, 3, 0, 0, 1), (32, 32, 23040, 1, 0, 0),
])
def test_gemm_blas(be, M, N, K, tA, tB, pad):
    """tnt_gemm_blas_f32 (rocBLAS sgemm behind the C ABI, row-major operands with leading dimensions): same operand
    conventions and results as tnt_gemm_f32 without an epilogue; padding untouched; accumulate; run-to-run bit-exact."""
    rng = np.random.default_rng(M * 5 + N * 3 + K)
    A = rng.standard_normal((M, K))
    Bm = rng.standard_normal((K, N))
    As = A.T if tA else A
    Bs = Bm.T if tB else Bm
    lda, ldb, ldc = As.shape[1] + pad, Bs.shape[1] + pad, N + pad
    Ad = torch.zeros(As.shape[0], lda, device="cuda"); Ad[:, :As.shape[1]] = dev(As)
    Bd = torch.zeros(Bs.shape[0], ldb, device="cuda"); Bd[:, :Bs.shape[1]] = dev(Bs)
    Cd = torch.full((M, ldc), 7.0, device="cuda")
    be.gemm_blas(Ad, Bd, Cd, M, N, K, lda, ldb, ldc, bool(tA), bool(tB))
    close(Cd[:, :N], A @ Bm)
    if pad:
        assert (Cd[:, N:] == 7.0).all()
    C2 = torch.full((M, ldc), 7.0, device="cuda")
    be.gemm_blas(Ad, Bd, C2, M, N, K, lda, ldb, ldc, bool(tA), bool(tB))
    assert torch.equal(Cd, C2)                                   # atomics off: bitwise reproducible
    C3 = torch.ones(M, ldc, device="cuda")
    be.gemm_blas(Ad, Bd, C3, M, N, K, lda, ldb, ldc, bool(tA), bool(tB), accumulate=True)
    close(C3[:, :N], A @ Bm + 1.0)
    ref = torch.zeros(M, ldc, device="cuda")
    be.gemm(Ad, Bd, ref, M, N, K, lda, ldb, ldc, bool(tA), bool(tB))
    close(Cd[:, :N], ref[:, :N].cpu().numpy(), rtol=2e-5)


@pytest.mark.parametrize("M,N,K,tA,tB,pad,bias", [
    (960, 5001, 512, 0, 0, 3, True), (960, 512, 5001, 0, 1, 3, False), (512, 5001, 960, 1, 0, 3, False),
    (1024, 2048, 512, 0, 0, 0, True), (33, 17, 29, 0, 0, 0, True), (33, 17, 29, 0, 1, 2, False), (33, 17, 29, 1, 0, 1, False),
])
def test_gemm_lt(be, M, N, K, tA, tB, pad, bias):
    """tnt_gemm_lt_f32 (hipBLASLt, full-FP32 compute, optional bias epilogue): operand conventions and results of
    tnt_gemm_f32; float32-level accuracy against float64 (no reduced-precision passes); padding untouched; bitwise
    reproducible over repeated calls with the algorithm the first call picked."""
    rng = np.random.default_rng(M * 5 + N * 3 + K)
    A = rng.standard_normal((M, K))
    Bm = rng.standard_normal((K, N))
    bv = rng.standard_normal(N) if bias else None
    As = A.T if tA else A
    Bs = Bm.T if tB else Bm
    lda, ldb, ldc = As.shape[1] + pad, Bs.shape[1] + pad, N + pad
    Ad = torch.zeros(As.shape[0], lda, device="cuda"); Ad[:, :As.shape[1]] = dev(As)
    Bd = torch.zeros(Bs.shape[0], ldb, device="cuda"); Bd[:, :Bs.shape[1]] = dev(Bs)
    bd = dev(bv) if bias else None
    Cd = torch.full((M, ldc), 7.0, device="cuda")
    be.gemm_lt(Ad, Bd, Cd, M, N, K, lda, ldb, ldc, bool(tA), bool(tB), bias=bd)
    want = A @ Bm + (bv if bias else 0.0)
    close(Cd[:, :N], want, rtol=1e-5)                            # f32 sums over K; TF32-style passes would miss it by 100x
    if pad:
        assert (Cd[:, N:] == 7.0).all()
    for _ in range(5):
        C2 = torch.full((M, ldc), 7.0, device="cuda")
        be.gemm_lt(Ad, Bd, C2, M, N, K, lda, ldb, ldc, bool(tA), bool(tB), bias=bd)
        assert torch.equal(Cd, C2)


@pytest.mark.parametrize("rows,D,A", [(23040, 32, 32), (300, 32, 32), (129, 32, 32), (5, 32, 32)])
def test_attention_front_bwd(be, rows, D, A):
    """tnt_attention_front_bwd_f32 (LeakyReLU' + bias gradient + W1 gradient + dF contribution of the attention layer's
    hoisted Dense: one launch of row-chunk reducers + row updaters, one finalize) against float64; repeated launches
    bit-identical (fixed summation order)."""
    rng = np.random.default_rng(rows + D)
    pre, dP, F = rng.standard_normal((rows, A)), rng.standard_normal((rows, A)) * 0.1, rng.standard_normal((rows, D))
    W1, dF0 = rng.standard_normal((D, A)) * 0.3, rng.standard_normal((rows, D)) * 0.1
    g = dP * np.where(pre > 0, 1.0, 0.2)
    part = torch.zeros(be.attention_front_bwd_parts(rows, D, A), device="cuda")
    outs = []
    for rep in range(3):
        dF, dW1, db1 = dev(dF0), torch.full((D, A), 7.0, device="cuda"), torch.full((A,), 7.0, device="cuda")
        be.attention_front_bwd(dev(pre), dev(dP), dev(F), dev(W1), dF, dW1, db1, part, rows, D, A, 0.2)
        torch.cuda.synchronize()
        outs.append((dF, dW1, db1))
    dF, dW1, db1 = outs[0]
    close(dW1, F.T @ g, rtol=2e-5); close(db1, g.sum(0), rtol=2e-5); close(dF, dF0 + g @ W1.T, rtol=1e-5)
    for o in outs[1:]:
        assert all(torch.equal(x, y) for x, y in zip(outs[0], o))
    # with the feature Dropout' folded into the dF pass: same bits as the launch plus tnt_dropout_f32 on its result
    step_dev = torch.tensor([4], dtype=torch.int32, device="cuda")
    dFd, dW1d, db1d = dev(dF0), torch.zeros(D, A, device="cuda"), torch.zeros(A, device="cuda")
    be.attention_front_bwd(dev(pre), dev(dP), dev(F), dev(W1), dFd, dW1d, db1d, part, rows, D, A, 0.2,
                           drop=(0.25, 4242, 17, step_dev))
    ref = outs[0][0].clone()
    be.dropout(ref, ref, rows, D, D, 0, D, 0, 0.25, 4242, 17, 0, step_dev)
    assert torch.equal(dFd, ref) and torch.equal(dW1d, dW1) and torch.equal(db1d, db1)
    keep = keep_mask((rows, D), 0.25, 4242, 17, 4)
    assert np.array_equal(dFd.cpu().numpy() != 0, keep & (outs[0][0].cpu().numpy() != 0))


@pytest.mark.parametrize("T,B,C,rate,V", [(15, 64, 256, 0.2, 5001), (3, 5, 8, 0.0, 0), (7, 9, 36, 0.5, 33)])
def test_bias_act_drop_bwd_and_colsum_multi(be, T, B, C, rate, V):
    """tnt_bias_act_drop_bwd_f32 against the launches it replaces (dropout -> act_bwd -> colsum, + the riding colsum job),
    identical dropout pattern; tnt_colsum4_f32 against separate column sums, skipped jobs included."""
    rng = np.random.default_rng(T * B + C)
    rows = T * B
    dy, pre = rng.standard_normal((rows, C)), rng.standard_normal((rows, C))
    step_dev = torch.tensor([5], dtype=torch.int32, device="cuda")
    ref = dev(dy).clone()
    if rate > 0:
        be.dropout(ref, ref, rows, C, C, B, C, 0, rate, 77, 13, 0, step_dev)
    be.act_bwd(dev(pre), ref, ref, rows * C, 1, 0.2)
    db_ref = torch.zeros(C, device="cuda")
    be.colsum(ref, db_ref, rows, C, C, torch.zeros(4 * C + 64, device="cuda"))
    x1 = rng.standard_normal((rows, V + 3)) if V else None
    dx, db = dev(dy).clone(), torch.full((C,), 7.0, device="cuda")
    out1 = torch.full((max(V, 1),), 7.0, device="cuda")
    be.bias_act_drop_bwd(dx, dev(pre), dx, db, rows, C, C, 1, 0.2, B, C, 0, rate, 77, 13, step_dev,
                         extra=(dev(x1), out1, rows, V, V + 3) if V else None)
    assert torch.equal(dx, ref)
    close(db, db_ref.cpu().numpy(), rtol=1e-5)
    close(db, ref.cpu().double().numpy().sum(0), rtol=1e-5)
    if V:
        close(out1, x1[:, :V].sum(0), rtol=1e-5)
    # colsum_multi: jobs 0 and 2 used through the 4-slot entry, with a hole
    a_, b_ = rng.standard_normal((rows, C)), rng.standard_normal((B, 5))
    oa, ob = torch.full((C,), 7.0, device="cuda"), torch.full((5,), 7.0, device="cuda")
    ad, bd = dev(a_), dev(b_)
    be._call(be.lib.tnt_colsum4_f32, "tnt_colsum4_f32", ad.data_ptr(), oa.data_ptr(), rows, C, C, None, None, 0, 0, 0,
             bd.data_ptr(), ob.data_ptr(), B, 5, 5, None, None, 0, 0, 0, be._s())
    close(oa, a_.sum(0), rtol=1e-5); close(ob, b_.sum(0), rtol=1e-5)
    o1, o2, o3 = (torch.full((k,), 7.0, device="cuda") for k in (C, 5, 1))
    be.colsum_multi([(ad, o1, rows, C, C), (bd, o2, B, 5, 5), (bd.view(-1)[4:], o3, B, 1, 5)])
    close(o1, a_.sum(0), rtol=1e-5); close(o2, b_.sum(0), rtol=1e-5); close(o3, b_[:, 4:5].sum(0), rtol=1e-5)


@pytest.mark.parametrize("rows,C", [(23040, 32), (64, 512), (130, 12)])
def test_sync_batchnorm_pieces(be, rows, C):
    """tnt_batchnorm_stats / apply_stats / dx (the pieces of BatchNorm around the data-parallel collectives): two "replicas"
    holding the halves of one batch, their partials concatenated as an all-gather would, reproduce tnt_batchnorm_fwd/bwd of
    the whole batch (statistics, moving statistics, outputs, input gradient); one replica reproduces it trivially."""
    rng = np.random.default_rng(rows + C)
    x, dy = rng.standard_normal((2 * rows, C)) * 2 + 0.5, rng.standard_normal((2 * rows, C))
    gamma, beta = 1 + 0.1 * rng.standard_normal(C), 0.1 * rng.standard_normal(C)
    mm0, mv0 = 0.1 * rng.standard_normal(C), 1 + 0.1 * rng.random(C)
    f = lambda *sh: torch.zeros(*sh, device="cuda")
    nch = be.bn_nchunk(2 * rows)
    work = f(C * (2 * nch + 1))
    # whole batch, existing entry points
    mm_w, mv_w, y_w, xh_w, inv_w = dev(mm0), dev(mv0), f(2 * rows, C), f(2 * rows, C), f(max(2 * rows, C))
    be.batchnorm_fwd(dev(x), dev(gamma), dev(beta), mm_w, mv_w, y_w, xh_w, inv_w, 2 * rows, C, C, True, 1e-3, 0.99, work)
    dx_w, dg_w, db_w = f(2 * rows, C), f(C), f(C)
    be.batchnorm_bwd(dev(dy), xh_w, dev(gamma), inv_w, dx_w, dg_w, db_w, 2 * rows, C, C, True, work)
    # two replicas
    n1 = be.bn_nchunk(rows) * 2 * C
    allp = f(2 * n1)
    xs = [dev(x[:rows]), dev(x[rows:])]
    for r in range(2):
        be.batchnorm_stats(xs[r], rows, C, allp[r * n1:(r + 1) * n1])
    sums = f(2 * C)
    outs = []
    for r in range(2):
        mm, mv, y, xh, inv, mw = dev(mm0), dev(mv0), f(rows, C), f(rows, C), f(max(rows, C)), f(C)
        be.batchnorm_apply_stats(allp, 2, xs[r], dev(gamma), dev(beta), mm, mv, y, xh, inv, rows, C, C, 1e-3, 0.99, mw)
        dg, db = f(C), f(C)
        be.batchnorm_bwd(dev(dy[r * rows:(r + 1) * rows]), xh, dev(gamma), inv, None, dg, db, rows, C, C, True, work)
        sums[:C] += dg; sums[C:] += db
        outs.append((mm, mv, y, xh, inv))
    for r in range(2):
        mm, mv, y, xh, inv = outs[r]
        sl = slice(r * rows, (r + 1) * rows)
        close(mm, mm_w.cpu().numpy(), rtol=2e-6); close(mv, mv_w.cpu().numpy(), rtol=2e-5)
        close(y, y_w[sl].cpu().numpy(), rtol=2e-5); close(xh, xh_w[sl].cpu().numpy(), rtol=2e-5)
        close(inv[:C], inv_w[:C].cpu().numpy(), rtol=2e-5)
        dx = f(rows, C)
        be.batchnorm_dx(dev(dy[sl]), C, xh, dev(gamma), inv, sums[:C], sums[C:], dx, rows, C, 2 * rows)
        close(dx, dx_w[sl].cpu().numpy(), rtol=5e-5)
    close(sums[:C], dg_w.cpu().numpy(), rtol=2e-5); close(sums[C:], db_w.cpu().numpy(), rtol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("T,B,R", [(5, 8, 30), (15, 64, 360), (3, 5, 129)])
def test_dropout_metric_rider(be, T, B, R):
    """tnt_dropout_metric_f32 = tnt_dropout_f32 + the partials of tnt_attention_metric_f32 in one launch: both bit-identical
    to the separate launches (time-major logical layout of the head's Dropout, lc_NIC.py:259)."""
    rng = np.random.default_rng(T + B + R)
    H = 64
    x = dev(rng.standard_normal((T * B, H)))
    alpha = dev(O.softmax(rng.standard_normal((T, B, R)), axis=-1))
    step_dev = torch.tensor([5], dtype=torch.int32, device="cuda")
    y0, y1 = torch.zeros_like(x), torch.full_like(x, float("nan"))
    npart = be.attention_metric_parts(T, R)
    p0, p1 = torch.zeros(npart, device="cuda"), torch.full((npart,), float("nan"), device="cuda")
    be.dropout(x, y0, T * B, H, H, B, H, 0, 0.4, 99, 7, 0, step_dev)
    be.attention_metric(alpha, None, p0, T, B, R)
    be.dropout_metric(x, y1, T * B, H, H, B, H, 0, 0.4, 99, 7, 0, step_dev, alpha, p1, T, B, R)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(p0, p1)

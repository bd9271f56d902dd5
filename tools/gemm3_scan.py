"""Times every configuration of tnt_gemm3_f32 (csrc/gemm3.hip) on the hot-path GEMM shapes next to the vendor kernels the
step used in round 2 (tnt_gemm_lt_f32 for the vocabulary-sized shapes, tnt_gemm_blas_f32 for the LSTM-sized ones) and checks
every result against torch.matmul in float64.  Interleaved rounds in ONE process (the guide's rule 24): each round times every
candidate once (REPS back-to-back launches), the table reports the median over rounds."""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import masters_thesis_amd.ops as ops
be = ops.backend()
# candidates: tile[:splitk] ...  (tiles: table of tnt_gemm3_f32 in csrc/gemm3.hip)
CFGS = [tuple(int(v) for v in c.split(":")) if ":" in c else (int(c), 1)
        for c in os.environ.get("G3_CFGS", "1,2,3,4,5,6,7,8,1:2,2:2,3:2,3:4,8:2,6:2,1:10,3:8,3:16,6:4").split(",")]
ROUNDS, REPS = int(os.environ.get("G3_ROUNDS", "5")), int(os.environ.get("G3_REPS", "20"))
shapes = [  # name, M, N, K, tA, tB, bias
    ("head fwd NN", 960, 5001, 512, 0, 0, 1), ("head dW TN", 512, 5001, 960, 1, 0, 0), ("head dX NT", 960, 512, 5001, 0, 1, 0),
    ("xproj NN", 1024, 2048, 512, 0, 0, 0), ("dU TN", 512, 2048, 1024, 1, 0, 0), ("dXin NT", 1024, 512, 2048, 0, 1, 0),
    ("c3 head fwd NN", 960, 5001, 256, 0, 0, 1), ("c3 head dW TN", 256, 5001, 960, 1, 0, 0), ("c3 head dX NT", 960, 256, 5001, 0, 1, 0),
    ("c3 inter fwd NN", 960, 256, 512, 0, 0, 1), ("c3 inter dW TN", 512, 256, 960, 1, 0, 0), ("c3 inter dX NT", 960, 512, 256, 0, 1, 0),
    ("c3 xproj NN", 960, 2048, 544, 0, 0, 0), ("c3 dW TN", 544, 2048, 960, 1, 0, 0), ("c3 dXin NT", 960, 544, 2048, 0, 1, 0),
]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(a in s[0] for a in sys.argv[1:])]
if os.environ.get("G3_KSCALE"):      # the same output at 2x and 4x the depth: time = fixed + per-stage cost
    shapes = [(f"{n} K*{f}", M, N, K * f, tA, tB, hb) for (n, M, N, K, tA, tB, hb) in shapes for f in (1, 2, 4)]


def timeit(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


r4 = lambda n: (n + 3) // 4 * 4
for name, M, N, K, tA, tB, hb in shapes:
    lda = r4(M if tA else K); ldb = r4(K if tB else N); ldc = r4(N)
    A = torch.zeros((K if tA else M), lda, device="cuda"); A[:, :(M if tA else K)].normal_()
    Bm = torch.zeros((N if tB else K), ldb, device="cuda"); Bm[:, :(K if tB else N)].normal_()
    bias = torch.randn(ldc, device="cuda") if hb else None
    Cm = torch.zeros(M, ldc, device="cuda")
    opA = (A[:, :M].t() if tA else A[:, :K]).double()
    opB = (Bm[:, :K].t() if tB else Bm[:, :N]).double()
    want = opA @ opB + (bias[:N].double() if hb else 0.0)
    scale = want.abs().max().item()
    cands = {}
    if max(N, K) >= 4096:
        cands["lt"] = lambda: be.gemm_lt(A, Bm, Cm, M, N, K, lda, ldb, ldc, transA=bool(tA), transB=bool(tB), bias=bias)
    if K < 4096 and N < 4096:
        sk = 1
        while sk * 2 * ((M + 63) // 64) * ((N + 63) // 64) <= 1280 and K // (sk * 2) >= 128 and sk < 64 and K >= 256: sk *= 2
        wk = torch.empty(sk * M * N, device="cuda")
        cands["old"] = lambda: be.gemm(A, Bm, Cm, M, N, K, lda, ldb, ldc, transA=bool(tA), transB=bool(tB), bias=bias, splitk=sk, work=wk)
    if not hb:
        cands["blas"] = lambda: be.gemm_blas(A, Bm, Cm, M, N, K, lda, ldb, ldc, transA=bool(tA), transB=bool(tB))
    dual = bool(tA and not tB and os.environ.get("G3_DUAL"))          # TN: second product + column sums ride along
    if dual:
        A2 = torch.zeros_like(A); A2[:, :M].normal_()
        C2 = torch.zeros(M, ldc, device="cuda"); col = torch.zeros(ldc, device="cuda")
        want2 = A2[:, :M].t().double() @ opB
        wantc = opB.sum(0)
    pt, ps = be.gemm3_plan(M, N, K, bool(tA), bool(tB))
    plan_key = f"g3/{pt}:{ps}"
    for tile, sk in CFGS + ([(pt, ps)] if (pt, ps) not in CFGS else []):
        if dual and sk == 1:
            def run(tile=tile):
                be.gemm3(A, Bm, Cm, M, N, K, lda, ldb, ldc, transA=True, transB=False, colsum=col, A2=A2, C2=C2, tile=tile)
            run.dual = True
            cands[f"g3x2/{tile}:1"] = run
        wf = be.gemm3_work_floats(M, N, tile, sk)
        if sk > 1 and wf <= 0:
            continue
        work = torch.empty((max(wf, 4) + 3) // 4 * 4, device="cuda") if sk > 1 else None
        if work is not None: be.gemm3_work_arm(work)
        sync = torch.zeros(be.gemm3_sync_words(M, N, tile) + 1, dtype=torch.int32, device="cuda") if sk > 1 else None
        cands[f"g3/{tile}:{sk}"] = (lambda tile=tile, sk=sk, work=work, sync=sync: be.gemm3(
            A, Bm, Cm, M, N, K, lda, ldb, ldc, transA=bool(tA), transB=bool(tB), bias=bias, tile=tile, splitk=sk, work=work, sync=sync))
        cands[f"g3/{tile}:{sk}"].sync = sync
    status, times = {}, {k: [] for k in cands}
    for k, fn in list(cands.items()):
        Cm.zero_()
        try:
            fn(); fn()
            torch.cuda.synchronize()
        except Exception as e:
            status[k] = "n/a"; del cands[k]; continue
        err = (Cm[:, :N].double() - want).abs().max().item() / scale
        if getattr(fn, "dual", False):
            err = max(err, (C2[:, :N].double() - want2).abs().max().item() / want2.abs().max().item(),
                      (col[:N].double() - wantc).abs().max().item() / wantc.abs().max().item())
        pad_clean = bool((Cm[:, N:] == 0).all().item())
        sy = getattr(fn, "sync", None)
        sync_ok = sy is None or not bool(sy.any().item())
        status[k] = ("ok" if err < 2e-6 * max(1.0, (K / 1024) ** 0.5) * 2 and pad_clean and sync_ok
                     else f"ERR {err:.1e}{'' if pad_clean else ' pad'}{'' if sync_ok else ' sync'}")
    for _ in range(ROUNDS):
        for k, fn in cands.items():
            times[k].append(timeit(fn, REPS))
    fl = 2.0 * M * N * K
    res = sorted((statistics.median(v), min(v), k) for k, v in times.items() if v)
    tf = lambda k, u: fl * (2 if k.startswith("g3x2") else 1) / u / 1e6
    print(f"{name:15s} {M}x{N}x{K} plan={plan_key} | " + " ".join(f"[{k} {u:.1f}us(min {mn:.1f}) {tf(k, u):.0f}TF {status[k]}]" for u, mn, k in res), flush=True)

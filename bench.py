#!/usr/bin/env python3
"""Benchmark: caption-tokens/sec, training (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dense|attention]

One "step" = one full train_step (forward + backward + clip + Adam) of the hot path on one
synthetic batch already resident in HBM.  Workload at N=1 is BASELINE config 2
(20k-voxel dense encoder + 512-unit LSTM, B=64, T=15, V=5001); config 3 (region-wise encoder +
attention) is timed in the same run and reported as the ``config3`` sub-object (``--workload
attention`` makes it the headline instead).  N>1: one process per GPU, weak scaling (B=64 per
GPU), gradient exchange over RCCL; when WORLD_SIZE is not set the script starts the N ranks itself
(``python -m torch.distributed.run``, before anything touches the GPU) and relays rank 0's line.
Rank 0 prints ONE JSON line.

``roofline`` describes the C-ABI call group that takes the most device time per step, found live:
one eager step is recorded (every launch goes through HipBackend._call), each distinct call is then
timed back to back with HIP events on the launch stream, and its algorithmic work (SURVEY 8d
formulas at the run's dims) is priced against the gfx950 peak that bounds it.  ``kernels`` lists the
other groups, ``step_frac`` prices the whole step (30.3 GFLOP for config 2) against the FP32-MFMA peak.
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # (the runtime's default here; explicit: masters-thesis_amd/__init__.py)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

B, T, V, U, E, N_VOX = 64, 15, 5001, 512, 512, 20000
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: FP32 matrix peak
# HBM bytes per launch from rocprofv3 --pmc passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes), by call
# group; filled from the summaries committed under profiles/ (None = not collected for that kernel)
_PMC3 = "profiles/r03_pmc_gemm3_and_chains.txt"
_PMC3C = "profiles/r03_pmc_chains_end_of_round.txt"         # the chain kernels as they stand at the end of the round
_PMC3N = "profiles/r03_pmc_end_of_round_nt.txt"             # the optimizer kernels with non-temporal moments (end of round 3)
_kib = lambda fetch, write: int((2 * fetch + write) * 1024)
PMC_TRAFFIC = {
    "dense": {"tnt_gemm3_pair_f32 TN 512x5001x960 + NT 960x512x5001": (_kib(69608.1, 23574.0), _PMC3),
              "tnt_gemm3_pair_f32 TN 512x2048x1024 x2 + NT 1024x512x2048": (_kib(54473.9, 14344.0), _PMC3),
              "tnt_gemm3_f32 NN 960x5001x512": (_kib(13164.1, 18885.0), _PMC3),
              "tnt_gemm3_f32 NN 1024x2048x512": (_kib(10295.9, 8192.0), _PMC3),
              "tnt_lstm_seq_fwd_f32 S=16 B=64 U=512": (_kib(21653.1, 14239.1), _PMC3C),
              "tnt_lstm_seq_bwd_f32 S=16 B=64 U=512": (_kib(28751.2, 23211.6), _PMC3C),
              "tnt_dense_dw_adam_f32 20000x512x64": (_kib(66692.6, 120000.0), _PMC3N),
              "tnt_dense_dw_adam_fin_f32 20000x512x64": (_kib(66692.6, 120000.0), _PMC3N),     # the same kernel and operands
              "tnt_adam_fin_f32": (_kib(56762.8, 84745.7), _PMC3N)},
    "attention": {"tnt_lc_seq_fwd_drop_f32 T=15 B=64 R=360 U=512": (_kib(27245.3, 15566.9), _PMC3C),
                  "tnt_lc_seq_bwd_drop_f32 T=15 B=64 R=360 U=512": (_kib(32031.3, 39104.4), _PMC3C),
                  "tnt_gemm3_pair_f32 TN 256x5001x960 + NT 960x256x5001": (_kib(62224.7, 21897.9), _PMC3),
                  "tnt_gemm3_pair_f32 TN 512x2048x960 x2 + NT 960x512x2048": (_kib(51075.9, 13960.0), _PMC3),
                  "tnt_gemm3_f32 NN 960x5001x256": (_kib(6616.8, 18885.0), _PMC3),
                  "tnt_adam_fin_f32": (_kib(53628.8, 79882.9), _PMC3N)},
}
STEP_GFLOP = {"dense": 30.3, "attention": 22.0}          # SURVEY 8d, whole training step
WORKLOAD_NAME = {
    "dense": "config 2: AttemptFour NIC.py dense 20000->512 encoder + BatchNorm + 512-unit LSTM, V=5001, T=15, B=64/GPU",
    "attention": "config 3: lc_NIC locally-dense 20000->360x32 + additive attention + 512-unit LSTM, V=5001, T=15, B=64/GPU",
}


def synth(rank, device):
    """SURVEY 8d synthetic batch: betas ~ N(0,1); caption = <start>, 6..13 tokens, <end>, zero pad."""
    import numpy as np
    import torch
    rng = np.random.default_rng(42 + rank)
    x = rng.standard_normal((B, N_VOX)).astype(np.float32)
    cap = np.zeros((B, T), np.int32)
    for b in range(B):
        L = int(rng.integers(6, 14))
        cap[b, 0] = 1
        cap[b, 1:1 + L] = rng.integers(3, V, size=L)
        cap[b, 1 + L] = 2
    tgt = np.zeros_like(cap)
    tgt[:, :-1] = cap[:, 1:]
    z = np.zeros((B, U), np.float32)
    to = lambda a, dt: torch.as_tensor(a, dtype=dt).to(device)
    return ((to(x, torch.float32), to(cap, torch.int32), to(z, torch.float32), to(z, torch.float32)),
            to(tgt, torch.int32)), (x, cap, z, tgt)


def make_model(workload, device, grad_sync=None):
    from masters_thesis_amd.optimizers import Adam
    if workload == "dense":
        from masters_thesis_amd.nic import NIC
        # dropout rates of AttemptFour/config.yaml:36-41 mapped onto NIC.py's three rates
        model = NIC(N_VOX, U, E, V, T, 0.0, 0.2, 0.2, 0.01, 0.00003, 0.00001, device=device, seed=42,
                    grad_sync=grad_sync)
    else:
        from masters_thesis_amd.lc_nic import NIC as LcNIC, synthetic_groups
        groups = synthetic_groups(N_VOX, 360, 32, seed=42)
        model = LcNIC(groups, U, 512, E, 32, V, T, 0.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.01, 0.001, 0.00003, 0.00001,
                      device=device, seed=42, grad_sync=grad_sync)
    model.compile(Adam(learning_rate=0.0001, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    return model


def cpu_baseline(workload, host_batch, budget_s=20.0):
    """The oracle (numpy float32 port of the reference path) timed on this host's cores."""
    import numpy as np
    from oracle import models as M
    x, cap, z, tgt = host_batch
    rng = np.random.default_rng(0)
    if workload == "dense":
        orc = M.NICDense(N_VOX, U, E, V, T, 0.0, 0.2, 0.2, 0.01, 3e-5, 1e-5).init_params(rng, np.float32)
    else:
        from masters_thesis_amd.lc_nic import synthetic_groups
        orc = M.LcNIC(synthetic_groups(N_VOX, 360, 32, seed=42), U, 512, E, 32, V, T, 0.0, 0.2, 0.2, 0.2, 0.2, 0.2,
                      0.01, 0.001, 3e-5, 1e-5).init_params(rng, np.float32)
    opt = M.AdamState(orc.p, lr=1e-4, clipnorm=0.1)
    data = (x, cap, z, z)
    orc.train_step(data, tgt, opt, M.DropCtx(seed=42, step=0, training=True))     # warm-up

    def timed(budget):
        n, t0 = 0, time.perf_counter()
        while True:
            orc.train_step(data, tgt, opt, M.DropCtx(seed=42, step=n + 1, training=True))
            n += 1
            el = time.perf_counter() - t0
            if el > budget or n >= 50:
                return n, el
    try:
        import threadpoolctl
        cores = max(i.get("num_threads", 1) for i in threadpoolctl.threadpool_info() if i.get("user_api") == "blas")
    except Exception:
        threadpoolctl, cores = None, os.cpu_count()
    n, el = timed(budget_s * 0.5)
    rates = {int(cores): (n * B * T / el, n, el)}
    if threadpoolctl is not None and cores > 8:        # SURVEY 8d also asks for the 8-thread figure
        with threadpoolctl.threadpool_limits(limits=8, user_api="blas"):
            n8, el8 = timed(budget_s * 0.5)
        rates[8] = (n8 * B * T / el8, n8, el8)
    best = max(rates, key=lambda c: rates[c][0])        # the small GEMMs of this step oversubscribe a 64-thread BLAS
    v, nb, elb = rates[best]
    out = {"value": round(v, 1), "unit": "caption-tokens/s", "cores": best, "kind": "port",
           "sample": f"{nb} full train steps of the same workload (numpy float32 oracle, {best} BLAS threads), {elb:.1f} s",
           "by_threads": {str(c): round(r[0], 1) for c, r in sorted(rates.items())}}
    return out


# ------------------------------------------------------------------------------------------ live per-kernel roofline
def _work_model(name, a):
    """(group key, algorithmic FLOPs, algorithmic HBM bytes) of one recorded C-ABI call (argument layouts: include/tnt_hip.h).
    SURVEY 8d: a GEMM is 2*M*N*K FLOPs and reads A, B / writes C once; an LSTM step is 2*B*4U*U FLOPs; the optimizer
    moves 7 words per parameter, the norm pass 2."""
    if name == "tnt_gemm3_pair_f32":        # two products in one launch: (address of tnt_gemm3_desc) x 2
        from masters_thesis_amd.ops import _G3Desc
        keys, fl, by = [], 0.0, 0.0
        for addr in a[:2]:
            d = _G3Desc.from_address(addr)
            nb = 2 if d.A2 else 1
            keys.append(("T" if d.transA else "N") + ("T" if d.transB else "N") + f" {d.M}x{d.N}x{d.K}" + (" x2" if nb == 2 else ""))
            fl += 2.0 * d.M * d.N * d.K * nb
            by += 4.0 * (nb * d.M * d.K + d.K * d.N + nb * d.M * d.N)
        return f"{name} " + " + ".join(keys), fl, by
    if name in ("tnt_gemm_f32", "tnt_gemm_blas_f32", "tnt_gemm_fused_f32", "tnt_gemm_lt_f32", "tnt_gemm3_f32"):
        nb = 1
        if name == "tnt_gemm3_f32":         # (A, B, C, bias, colsum, A2, C2, M, N, K, lda, ldb, ldc, tA, tB, tile, splitk, ...)
            M, N, K, tA, tB = a[7], a[8], a[9], a[13], a[14]
            nb = 2 if a[5] else 1
        elif name == "tnt_gemm_lt_f32":
            M, N, K, tA, tB = a[4], a[5], a[6], a[10], a[11]
        elif name == "tnt_gemm_f32":
            M, N, K, tA, tB = a[5], a[6], a[7], a[11], a[12]
        elif name == "tnt_gemm_fused_f32":
            M, N, K, tA, tB = a[7], a[8], a[9], a[13], a[14]
            nb = 2 if a[5] else 1
        else:
            M, N, K, tA, tB = a[3], a[4], a[5], a[9], a[10]
        lay = ("T" if tA else "N") + ("T" if tB else "N")
        return (f"{name} {lay} {M}x{N}x{K}" + (" x2" if nb == 2 else ""), 2.0 * M * N * K * nb,
                4.0 * (nb * M * K + K * N + nb * M * N))
    if name == "tnt_lstm_step_bwd_f32":
        Bq, Uq = a[17], a[18]
        if not a[0]:
            return f"{name} (last step, no matmul)", 0.0, 0.0
        return f"{name} B={Bq} U={Uq}", 2.0 * Bq * 4 * Uq * Uq, 0.0
    if name == "tnt_lstm_step_fwd_f32":
        Bq, Uq, D = a[15], a[16], a[6]
        return f"{name} B={Bq} U={Uq} D={D}", 2.0 * Bq * 4 * Uq * (Uq + D), 0.0
    if name in ("tnt_lstm_seq_fwd_f32", "tnt_lstm_seq_bwd_f32"):
        S, Bq, Uq = a[10], a[11], a[12]
        return f"{name} S={S} B={Bq} U={Uq}", 2.0 * S * Bq * 4 * Uq * Uq, 0.0
    if name in ("tnt_lc_seq_fwd_f32", "tnt_lc_seq_bwd_f32", "tnt_lc_seq_fwd_drop_f32", "tnt_lc_seq_bwd_drop_f32"):
        # config 3's chains: LSTM step + attention step, T times (the _drop forms carry the LSTM-output Dropout as a
        # rider; T, B, R, D, A, U sit at the same argument positions in all four, include/tnt_hip.h)
        Tq, Bq, R, D, A, Uq = a[19:25]
        fwd = "_fwd_" in name
        lstm = 2.0 * Bq * 4 * Uq * (Uq + D)
        att = (2.0 * Bq * Uq * A + 2.0 * Bq * R * (A + D)) * (1 if fwd else 2)
        return f"{name} T={Tq} B={Bq} R={R} U={Uq}", Tq * (lstm + att), 0.0
    if name == "tnt_dense_dw_skinny_f32":
        Nq, Eq, Bk = a[3], a[4], a[5]
        return f"{name} {Nq}x{Eq}x{Bk}", 2.0 * Nq * Eq * Bk, 4.0 * (Nq * Eq + Bk * Nq + Bk * Eq)
    if name in ("tnt_dense_fwd_stream_f32", "tnt_dense_fwd_stream_gram_f32"):       # streams the kernel once, writes nsplit partials
        o = 2 if name.endswith("gram_f32") else 0                                       # (+ gx_part, w2_part pointers)
        Bq, Eq, K, ns = a[3 + o], a[4 + o], a[5 + o], a[8 + o]
        return f"{name} {Bq}x{Eq}x{K}", 2.0 * Bq * Eq * K, 4.0 * (K * Eq + Bq * K + ns * Bq * Eq)
    if name == "tnt_dense_dw_sqnorm_f32":        # the skinny product again, reading theta (norm of g + 2 l2 theta)
        Nq, Eq, Bk = a[6], a[7], a[8]
        return f"{name} {Nq}x{Eq}x{Bk}", 2.0 * Nq * Eq * Bk, 4.0 * (Nq * Eq + Bk * Nq + Bk * Eq)
    if name == "tnt_dense_dw_adam_f32":          # ... and clip + Adam on it: theta, m, v read and written, g never stored
        Nq, Eq, Bk = a[14], a[15], a[16]
        return f"{name} {Nq}x{Eq}x{Bk}", 2.0 * Nq * Eq * Bk, 4.0 * (6 * Nq * Eq + Bk * Nq + Bk * Eq)
    if name == "tnt_dense_dw_adam_fin_f32":      # the same launch without a finalize launch in front (partial, k0, k1 replace sq)
        Nq, Eq, Bk = a[16], a[17], a[18]
        return f"{name} {Nq}x{Eq}x{Bk}", 2.0 * Nq * Eq * Bk, 4.0 * (6 * Nq * Eq + Bk * Nq + Bk * Eq)
    if name == "tnt_softmax_cce_f32":
        rows, ld = a[6], a[8]
        return f"{name} {rows}x{a[7]}", 0.0, 8.0 * rows * ld
    return name, None, None


def kernel_breakdown(model, batch, workload, reps=20, limit=12):
    """Records the launches of ONE eager training step, times every distinct C-ABI call back to back with HIP events
    on the launch stream, and returns (dominant group's roofline dict, [other groups, largest first])."""
    import torch
    be = model.be
    saved_graphs = model._graphs
    model._graphs = {}                          # next train_step runs its launch sequence eagerly
    # the recorded calls are re-issued many times below: keep the metrics-ring job (which advances a device counter the host
    # mirrors) out of the recording, and leave the model's notion of which captured steps file their metrics there untouched
    saved_ring = (model.__dict__.get("metric_ring", None), set(model.__dict__.get("_ring_keys", ())))
    model.metric_ring = False
    be._rec = rec = []
    try:
        model.train_step(batch)
    finally:
        be._rec = None
        if saved_ring[0] is None:
            del model.metric_ring
        else:
            model.metric_ring = saved_ring[0]
        model._ring_keys = saved_ring[1]
    torch.cuda.synchronize()
    model._graphs = saved_graphs
    n_param = int(model.arena.total)
    if getattr(model, "_enc_last_fused", None):       # the encoder kernel is updated by tnt_dense_dw_adam_f32, not by these
        n_param -= int(model.arena.entries["dense_img/kernel"].size)
    groups = {}
    for fn, name, args in rec:
        key, fl, by = _work_model(name, args)
        if name in ("tnt_adam_f32", "tnt_adam_ring_f32", "tnt_adam_fin_f32"):
            fl, by = 0.0, 28.0 * n_param
        elif name in ("tnt_seg_sqnorm_f32", "tnt_span_sqnorm_f32", "tnt_span_sqnorm_lr_f32"):
            fl, by = 0.0, 8.0 * n_param
        g = groups.setdefault(key, {"calls": 0, "fn": fn, "args": args, "flops": fl, "bytes": by})
        g["calls"] += 1
    out = []
    for key, g in groups.items():
        fn, args = g["fn"], g["args"]
        for _ in range(2):
            fn(*args)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            fn(*args)
        ev1.record()
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 1e3 / reps
        row = {"kernel": key, "calls_per_step": g["calls"], "avg_launch_us": round(us, 2),
               "us_per_step": round(us * g["calls"], 1)}
        fl, by = g["flops"], g["bytes"]
        if fl is not None and (fl > 0 or by > 0):
            t_f = fl / (MFMA_F32_PEAK_TF * 1e12)
            t_b = by / (HBM_PEAK_GBS * 1e9)
            if t_f >= t_b:
                ach = fl / (us * 1e-6) / 1e12
                row.update(bound="mfma", achieved=round(ach, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                           frac=round(ach / MFMA_F32_PEAK_TF, 4), flops_per_launch=fl)
            else:
                ach = by / (us * 1e-6) / 1e9
                row.update(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                           frac=round(ach / HBM_PEAK_GBS, 4), bytes_per_launch=by)
            pmc = PMC_TRAFFIC.get(workload, {}).get(key)
            row["traffic"] = pmc[0] if pmc else None
            if pmc:
                row["traffic_source"] = pmc[1]
        out.append(row)
    out.sort(key=lambda r: -r["us_per_step"])
    dom = out[0]                       # the call group with the most device time per step, whatever it is
    if "bound" not in dom:
        raise RuntimeError(f"bench.py: the dominant call group '{dom['kernel']}' ({dom['us_per_step']} us/step) has no "
                           "work model in _work_model(); price it before reporting a roofline")
    others = out[1:] if limit is None else out[1:1 + limit]
    total = round(sum(r["us_per_step"] for r in out), 1)
    return dom, others, total


def step_percentiles(model, batch, steps):
    """p10 / p50 / p90 of the per-step device time (HIP events between consecutive steps), measured in a
    separate loop after the timed region so the K timed steps stay undisturbed."""
    import torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(steps):
        model.train_step(batch)
        ev[i + 1].record()
    torch.cuda.synchronize()
    d = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    pick = lambda q: round(d[min(steps - 1, int(q * steps))], 4)
    return [pick(0.10), pick(0.50), pick(0.90)]


SETTLE_STEPS = 30      # untimed steps in front of the W warm-up steps: the first calls build buffers, record the step's hipGraph and
                       # raise the clocks; with W = 5 alone the K timed steps measured 0.513 ms/step, with these 0.493 (W = 50: 0.493)


def timed_steps(model, batch, steps, warmup, world=1, dist=None, device=None):
    """SETTLE_STEPS untimed set-up steps, W warm-up steps, then exactly K steps between barrier + synchronize on both sides;
    max over ranks."""
    import torch
    for _ in range(max(0, SETTLE_STEPS - warmup)):
        model.train_step(batch)
    for _ in range(warmup):
        model.train_step(batch)
    torch.cuda.synchronize()
    # The persistent LSTM kernel guards its barriers.  If the guard ever trips during the warm-up (it never has), every
    # rank goes back to the per-step kernels together and warms up again, so the timed region measures valid steps.
    sync = getattr(model, "seq_sync", None)
    tripped = torch.tensor([1 if (sync is not None and int(sync[1024].item()) != 0) else 0], dtype=torch.int32, device=device)
    if world > 1:
        dist.all_reduce(tripped, op=dist.ReduceOp.MAX)
    note = None
    if int(tripped.item()):
        model.disable_seq_lstm()
        note = "persistent LSTM kernel disabled after a guard trip in warm-up; per-step kernels measured"
        for _ in range(max(3, warmup)):
            model.train_step(batch)
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.train_step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el, note


def dp_world1_rehearsal(args, device):
    """ms/step of the data-parallel schedule at world size 1 over RCCL (both workloads), next to which the single-process
    ms_per_step shows what the schedule itself costs (segments, unfused update, dense embedding backward) before any
    communication.  Run in a child process (it needs a process group; this process keeps its single-GPU state)."""
    out = {}
    for name, wl, extra in (("dense", "dense", {}), ("attention", "attention", {}),
                            ("dense_row_sharded_encoder", "dense", {"TNT_DP_SHARD_ENC": "1"})):      # opt-in form (dp.py)
        cmd = [sys.executable, os.path.abspath(__file__), "--force-dp", "--workload", wl, "--steps", str(min(args.steps, 200)),
               "--warmup", str(min(args.warmup, 20)), "--no-cpu-baseline", "--no-config3", "--no-dp-world1"]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env["MASTER_PORT"] = str(29600 + os.getpid() % 300)
        env.update(extra)
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        out[name] = round(json.loads(lines[-1])["ms_per_step"], 4) if p.returncode == 0 and lines else None
    return {"ms_per_step": out, "note": "dp.attach(model, 1) over RCCL on this GPU: schedule overhead only, no scaling claim"}


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks with torch.distributed.run as a CHILD process (this process has not
    touched the GPU and never will) and relay rank 0's JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if p.returncode != 0 or not lines:
        sys.stdout.write(p.stdout)
        raise SystemExit(p.returncode or 1)
    print(lines[-1], flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="dense", choices=["dense", "attention"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config3", action="store_true", help="skip the config-3 sub-measurement of the default run")
    ap.add_argument("--host-inputs", action="store_true", help="feed numpy batches (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--force-dp", action="store_true", help="run the data-parallel schedule even at world size 1 (rehearsal)")
    ap.add_argument("--no-dp-world1", action="store_true", help="skip the world-size-1 rehearsal of the data-parallel schedule")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dp = world > 1 or args.force_dp
    dist = None
    if use_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints banner lines ("Hostname", "Librccl path") on fd 1 when the communicator is created;
        # stdout must carry exactly one JSON line, so fd 1 points at stderr until the communicator exists.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    model = make_model(args.workload, device, None)
    if use_dp:
        from masters_thesis_amd import dp
        dp.attach(model, world, rank=rank)
        torch.cuda.synchronize()
        dist.barrier()
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    batch, host_batch = synth(rank, device)
    if args.host_inputs:
        x, cap, z, tgt = host_batch
        batch = ((x, cap, z, z), tgt)

    el, seq_note = timed_steps(model, batch, args.steps, args.warmup, world, dist, device)
    pct = step_percentiles(model, batch, min(args.steps, 200)) if args.steps >= 10 else None
    last = model.train_step(batch).as_floats()
    model.check_device_errors()          # a guard trip of the persistent LSTM kernel would void the run

    if rank == 0:
        tokens = world * B * T * args.steps
        ms = el / args.steps * 1e3
        out = {
            "metric": "caption-tokens/sec training (20k-voxel enc, 512 LSTM, B=64)",
            "value": round(tokens / el, 1), "unit": "caption-tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" (host numpy batches, PCIe-inclusive)" if args.host_inputs else ""),
            "config": {"workload": WORKLOAD_NAME[args.workload], "global_batch": B * world, "seq_len": T,
                       "parallelism": f"dp{world}", "final_loss": round(last["loss"], 4), "step_ms_p10_p50_p90": pct},
        }
        # how the step's launches are re-issued: recorded launch plans (ModelBase._run_planned) or a hipGraph
        plans = [k for k, v in model._graphs.items() if isinstance(v, tuple)]
        out["config"]["replay"] = ("launch plan" if plans and len(plans) == len(model._graphs) else
                                   "launch plans + hipGraph segments" if plans else "hipGraph")
        if seq_note:
            out["config"]["note"] = seq_note
        if not use_dp:
            dom, others, total = kernel_breakdown(model, batch, args.workload)
            out["roofline"] = dom
            out["kernels"] = others
            out["kernel_us_per_step"] = total
        gf = STEP_GFLOP[args.workload]
        out["step_frac"] = {"gflop_per_step": gf, "achieved_tflops": round(gf / ms, 2), "peak": MFMA_F32_PEAK_TF,
                            "frac": round(gf / ms / MFMA_F32_PEAK_TF, 4)}
        if world == 1 and not use_dp and args.workload == "dense" and not args.no_config3:
            del model
            torch.cuda.empty_cache()
            m3 = make_model("attention", device, None)
            k3 = min(args.steps, 200)
            el3, _ = timed_steps(m3, batch, k3, min(args.warmup, 20), 1, None, device)
            ms3 = el3 / k3 * 1e3
            dom3, others3, total3 = kernel_breakdown(m3, batch, "attention")
            m3.check_device_errors()
            out["config3"] = {"workload": WORKLOAD_NAME["attention"], "steps": k3, "ms_per_step": round(ms3, 4),
                              "value": round(B * T * k3 / el3, 1), "unit": "caption-tokens/s", "roofline": dom3,
                              "kernels": others3[:6], "kernel_us_per_step": total3,
                              "step_frac": round(STEP_GFLOP["attention"] / ms3 / MFMA_F32_PEAK_TF, 4)}
        if world == 1 and not use_dp and not args.no_dp_world1:
            out["dp_world1"] = dp_world1_rehearsal(args, device)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload, host_batch)
        print(json.dumps(out), flush=True)
    if use_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""what segment C1 of dp.PipelinedDenseSync costs on ONE GPU at the operand sizes of an 8-rank run --
every rank forms the encoder-kernel gradient dW[20000 x 512] = X_all^T dpre_all with K = G * B = 512 rows from the gathered
operands, takes its norm, clips and applies Adam -- next to the single-process form (K = 64, fused dW + clip + Adam, gradient
never written).  Prints a small table for profiles/."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
m = bench.make_model("dense", dev)
for _ in range(3):
    m.train_step(batch)
torch.cuda.synchronize()
be, a = m.be, m.arena
N, E = m.N, m.E


def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


sl = a.seg_slice(0, 1)                      # the encoder kernel's arena slice
print(f"{'G':>2s} {'K':>4s} {'dW us':>8s} {'norm+clip+Adam us':>18s} {'C1 total us':>12s}   product")
for G in (1, 2, 4, 8):
    K = 64 * G
    x_all = torch.randn(K, m.ldx, device=dev); dpre_all = torch.randn(K, E, device=dev) * 1e-3
    if K <= 64:
        prod, f = "tnt_dense_dw_skinny_f32", (lambda: be.dense_dw_skinny(x_all, dpre_all, a.g("dense_img/kernel"), N, E, K, m.ldx))
    else:
        prod, f = "tnt_gemm3_f32 TN (plan)", (lambda: m.gemm_sk(x_all, dpre_all, a.g("dense_img/kernel"), N, E, K, m.ldx, E, E, transA=True))
    t_dw = timeit(f)
    t_up = timeit(lambda: m._update_slice(sl))
    print(f"{G:2d} {K:4d} {t_dw:8.1f} {t_up:18.1f} {t_dw + t_up:12.1f}   {prod}")
m._enc_fused = None
t_f = timeit(lambda: m._bwd_enc(64, 15)) if False else None

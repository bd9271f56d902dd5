"""Time the streaming encoder forward (tnt_dense_fwd_stream_f32) + the partial-summing tail
against the generic split-K GEMM + tail, back-to-back launches on one stream (launch overhead included)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import masters_thesis_amd.ops as ops
be = ops.backend()
dev = torch.device("cuda", 0)
B, K, E, NS = 64, 20000, 512, int(os.environ.get("NS", "16"))
x = torch.randn(B, K, device=dev); w = torch.randn(K, E, device=dev) / 141; bias = torch.zeros(E, device=dev)
part = torch.zeros(NS * B * E, device=dev)
f = lambda *s: torch.zeros(*s, device=dev)
pre, out, xhat, inv, mm, mv, g, b = f(B, E), f(B, E), f(B, E), f(E), f(E), torch.ones(E, device=dev), torch.ones(E, device=dev), f(E)
step = torch.zeros(1, dtype=torch.int32, device=dev)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
s = lambda: be.dense_fwd_stream(x, w, part, B, E, K, K, E, NS)
tl = lambda: be.enc_tail_fwd_sk(part, NS, bias, pre, 0.2, g, b, mm, mv, out, xhat, inv, B, E, E, True, 1e-3, 0.99, 0.2, 0.2, 7, 2, 48, step)
print("NS", NS, "stream %.2f us  tail_sk %.2f us  both %.2f us" % (t(s), t(tl), t(lambda: (s(), tl()))))
dpre = torch.randn(B, E, device=dev) * 0.01; dw = torch.zeros(K, E, device=dev)
print("dense_dw_skinny %.2f us" % t(lambda: be.dense_dw_skinny(x, dpre, dw, K, E, B, K)))
theta = torch.randn(K, E, device=dev) * 0.05; m_ = torch.zeros(K, E, device=dev); v_ = torch.zeros(K, E, device=dev)
partial = torch.zeros(2 * 1250, device=dev); sq = torch.ones(1, device=dev); lrt = torch.full((1,), 1e-4, device=dev)
print("dense_dw_sqnorm %.2f us" % t(lambda: be.dense_dw_sqnorm(x, dpre, theta, 0.01, partial, 1250, K, E, B, K)))
print("dense_dw_adam %.2f us" % t(lambda: be.dense_dw_adam(x, dpre, theta, m_, v_, 0.01, sq, None, lrt, 0.9, 0.98, 1e-8, 0.1, K, E, B, K)))

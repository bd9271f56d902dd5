"""Runs the dense-encoder kernels of a config-2 step a few times (for rocprofv3 --pmc passes): streaming forward with the
Gram by-products, the fused clip + Adam pass over X^T dpre, the plain skinny dW, the Gram norm finish."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import masters_thesis_amd.ops as ops
be = ops.backend(); dev = torch.device("cuda", 0)
B, K, E, NS = 64, 20000, 512, 16
g = torch.Generator(device="cpu").manual_seed(3)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
x, w, bias, dpre = r(B, K), r(K, E) / 141, r(E) * 0.1, r(B, E) * 0.01
part, gx, w2 = torch.zeros(NS * B * E, device=dev), torch.zeros(NS * 64 * 64, device=dev), torch.zeros(NS * (E // 32), device=dev)
pre = torch.zeros(B, E, device=dev)
theta, m, v, dw = w.clone(), torch.zeros(K, E, device=dev), torch.zeros(K, E, device=dev), torch.zeros(K, E, device=dev)
partial = torch.zeros(2 * 1250, device=dev); sq = torch.ones(1, device=dev); lrt = torch.full((1,), 1e-4, device=dev)
for _ in range(10):
    be.dense_fwd_stream_gram(x, w, part, gx, w2, B, E, K, K, E, NS)
    be.dense_gram_norm(dpre, pre, bias, gx, NS, w2, NS * (E // 32), 0.01, partial, 1250, B, E)
    be.dense_dw_adam(x, dpre, theta, m, v, 0.01, sq, None, lrt, 0.9, 0.98, 1e-8, 0.1, K, E, B, K)
    be.dense_dw_skinny(x, dpre, dw, K, E, B, K)
torch.cuda.synchronize()

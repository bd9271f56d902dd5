"""lc_nic single-GPU step vs PipelinedAttentionSync at world 1: per-step, per-variable weight / gradient drift."""
import os, sys, socket
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.distributed as dist
import test_gpu_lcnic as T
from masters_thesis_amd import dp
from masters_thesis_amd.optimizers import Adam
dims = T.DIMS[1]
rates = (0.1, 0.2, 0.2, 0.2, 0.2, 0.2)
B, N, R, D, A, U, Et, V, Tn = dims
a, orc = T.build(np.random.default_rng(77), rates, dims)
b, _ = T.build(np.random.default_rng(77), rates, dims)
for k, v in orc.p.items():
    b.set_weight(k, v)
for m in (a, b):
    m.compile(Adam(learning_rate=1e-3, beta_1=0.9, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dp.attach(b, 1)
rng = np.random.default_rng(6)
for step in range(5):
    data, tgt = T.synth_batch(B, N, Tn, V, U, rng)
    a.train_step((data, tgt)); b.train_step((data, tgt))
    torch.cuda.synchronize()
    out = []
    for name, e in a.arena.entries.items():
        dth = (a.arena.p(name) - b.arena.p(name)).abs().max().item()
        dg = (a.arena.g(name) - b.arena.g(name)).abs().max().item()
        if dth > 1e-8 or dg > 1e-9:
            out.append(f"{name}: dtheta {dth:.1e} dgrad {dg:.1e} (|g| {a.arena.g(name).abs().max().item():.1e})")
    worst = max(a.arena.entries, key=lambda nm: (a.arena.p(nm) - b.arena.p(nm)).abs().max().item())
    print(step, "worst", worst, (a.arena.p(worst) - b.arena.p(worst)).abs().max().item(), "grad diff", (a.arena.g(worst) - b.arena.g(worst)).abs().max().item(), "whole", (a.arena.theta - b.arena.theta).abs().max().item())
    print(step, "sq diff", (a.arena.sq - b.arena.sq).abs().max().item(), out[:2], len(out))
dist.destroy_process_group()

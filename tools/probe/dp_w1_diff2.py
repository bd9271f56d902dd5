import os, sys, socket
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.distributed as dist
from test_gpu_nic import _twin_models, synth_batch
from masters_thesis_amd import dp
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
a, b, (B, N, T, V, U, E) = _twin_models()
dp.attach(b, 1)
rng = np.random.default_rng(6)
data, tgt = synth_batch(B, N, T, V, U, rng)
a.train_step((data, tgt)).as_floats(); b.train_step((data, tgt)).as_floats()
torch.cuda.synchronize()
print("plans a", a._g3_plans); print("plans b", b._g3_plans)
for k, e in a.arena.entries.items():
    ga, gb = a.arena.grad[e.off:e.off + e.size], b.arena.grad[e.off:e.off + e.size]
    d = (ga - gb).abs().max().item()
    print(f"{k:40s} grad diff {d:.3e}  |g| {ga.abs().max().item():.3e}  sq a {a.arena.sq[e.seg].item():.9e} b {b.arena.sq[e.seg].item():.9e}")
for nm in ("logits", "Out", "dOut", "dZ", "dXin", "Hs"):
    print(nm, (getattr(a, nm) - getattr(b, nm)).abs().max().item())
dist.destroy_process_group()

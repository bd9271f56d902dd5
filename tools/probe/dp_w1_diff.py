"""debug: which variables differ between the single-GPU step and the world-1 DP schedule (tests/test_gpu_nic.py twin models)"""
import os, sys, socket
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.distributed as dist
from test_gpu_nic import _twin_models, synth_batch
from masters_thesis_amd import dp
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for g3, riders in ((True, True), (True, False), (False, False)):
    a, b, (B, N, T, V, U, E) = _twin_models()
    for m in (a, b):
        m.use_gemm3, m.g3_riders = g3, riders
    dp.attach(b, 1)
    rng = np.random.default_rng(6)
    for step in range(5):
        data, tgt = synth_batch(B, N, T, V, U, rng)
        a.train_step((data, tgt)).as_floats(); b.train_step((data, tgt)).as_floats()
        torch.cuda.synchronize()
        wa, wb = a.get_weights_dict(), b.get_weights_dict()
        worst = sorted(((np.abs(wa[k] - wb[k]).max(), k) for k in wa), reverse=True)[:3]
        print(g3, riders, step, [(f"{d:.2e}", k) for d, k in worst], flush=True)
dist.destroy_process_group()

"""Per-call breakdown (bench.kernel_breakdown) of the data-parallel schedule at world size 1 over RCCL, next to the
single-GPU step: which launches the schedule adds.  usage: dp_breakdown.py [dense|attention]"""
import os, sys
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from masters_thesis_amd import dp
wl = sys.argv[1] if len(sys.argv) > 1 else "dense"
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
batch, _ = bench.synth(0, dev)
for name, attach in (("single", False), ("dp world 1", True)):
    m = bench.make_model(wl, dev)
    if attach:
        dp.attach(m, 1, rank=0)
    for _ in range(20): m.train_step(batch)
    torch.cuda.synchronize()
    el, _ = bench.timed_steps(m, batch, 200, 3, 1, None, dev)
    dom, others, total = bench.kernel_breakdown(m, batch, wl, limit=None)
    rows = [dom] + others
    print(f"== {name}: step {el / 200 * 1e3:.4f} ms; {sum(r['calls_per_step'] for r in rows)} C-ABI launches, isolated sum {total} us")
    for k in rows:
        print(f"   {k['kernel'][:60]:60s} x{k['calls_per_step']:<3} {k['avg_launch_us']:7.2f} {k['us_per_step']:7.1f}")
dist.destroy_process_group()

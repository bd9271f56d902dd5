"""Every C-ABI call group of one training step, timed back to back (bench.kernel_breakdown without the top-12 cut).
usage: step_breakdown.py [dense|attention] [fused_finalize 0|1]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "dense"
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
m = bench.make_model(wl, dev)
if len(sys.argv) > 2: m.fused_finalize = bool(int(sys.argv[2]))
for _ in range(20): m.train_step(batch)
torch.cuda.synchronize()
el, _ = bench.timed_steps(m, batch, 300, 3, 1, None, dev)
dom, others, total = bench.kernel_breakdown(m, batch, wl, limit=None)
print(f"step {el / 300 * 1e3:.4f} ms; isolated launches sum {total} us")
for k in [dom] + others:
    print(f"{k['kernel'][:64]:64s} x{k['calls_per_step']:<3} {k['avg_launch_us']:7.2f} {k['us_per_step']:7.1f}  {k.get('frac', '')}")

"""N training steps of a bench workload for rocprofv3 --kernel-trace --stats (no breakdown, no CPU baseline):
usage: prof_step.py [dense|attention] [steps]   -> summarise with tools/prof_summary.py <dir> <steps>"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "dense"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
m = bench.make_model(wl, dev)
for _ in range(steps):
    m.train_step(batch)
torch.cuda.synchronize()

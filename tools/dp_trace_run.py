"""World-size-1 run of the data-parallel schedule for a rocprofv3 --kernel-trace (tools/trace_step.py reads it)."""
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
m = bench.make_model(wl, dev)
dp.attach(m, 1, rank=0)
for _ in range(40): m.train_step(batch)
torch.cuda.synchronize()
dist.destroy_process_group()

"""Host-side cost of one train_step call (enqueue only) against the device time per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
workload = "attention" if "--attention" in sys.argv else "dense"
model = bench.make_model(workload, dev, None)
if "--dp" in sys.argv:            # world-size-1 rehearsal of the data-parallel schedule (RCCL on one GPU)
    import os
    import torch.distributed as dist
    from masters_thesis_amd import dp
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dp.attach(model, 1)
batch, _ = bench.synth(0, dev)
for _ in range(30):
    model.train_step(batch)
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    model.train_step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.4f} ms/step; total {1e3 * (t2 - t0) / n:.4f} ms/step")
# pieces
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    model.train_step(batch)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

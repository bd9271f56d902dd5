"""Host-side cost of one train_step call (enqueue only) against the device time per step."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
model = bench.make_model("dense", dev, None)
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

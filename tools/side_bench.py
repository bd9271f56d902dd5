import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
for name, kw in (("no side streams", dict(use_side_streams=False)), ("side: seq only", dict(side_head=False)), ("side: head+seq", dict())):
    m = bench.make_model("dense", dev)
    for k, v in kw.items(): setattr(m, k, v)
    for _ in range(10): m.train_step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): m.train_step(batch)
    torch.cuda.synchronize(); print(f"{name:20s}: {(time.perf_counter() - t0) * 10:.4f} ms/step")

import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
for split in (True, False, True, False):
    m = bench.make_model("attention", dev)
    m.split_encoder = split
    for _ in range(20): m.train_step(batch)
    torch.cuda.synchronize()
    el, _ = bench.timed_steps(m, batch, 300, 3, 1, None, dev)
    print("split_encoder", split, f"{el / 300 * 1e3:.4f} ms/step")

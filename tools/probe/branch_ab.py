"""A/B: text branch and front branch of config 3's backward as parallel graph branches (branch_streams) or in sequence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
for on in (False, True, False, True):
    m = bench.make_model("attention", dev)
    m.use_side_streams = on
    m.branch_streams = on
    for _ in range(20): m.train_step(batch)
    torch.cuda.synchronize()
    el, _ = bench.timed_steps(m, batch, 300, 3, 1, None, dev)
    print("branch_streams", on, f"{el / 300 * 1e3:.4f} ms/step", m.train_step(batch).as_floats()["loss"])

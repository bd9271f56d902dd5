"""Runs the config-3 step with attribute overrides (for rocprofv3): python tools/att_variant.py r_lstm=0.0"""
import sys
import torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
model = bench.make_model("attention", dev, None)
for arg in sys.argv[1:]:
    k, v = arg.split("=")
    setattr(model, k, eval(v))
batch, _ = bench.synth(0, dev)
for _ in range(60):
    model.train_step(batch)
torch.cuda.synchronize()

"""A/B of the step time with the one-round fused gradient GEMMs switched on one by one (config 2)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
for name, flags in (("blas for all", dict(fused_lstm_grads=False)), ("lstm dW+dU+db fused", {}),
                    ("+ head dW+db fused", dict(fused_head_grads=True)), ("+ xproj", dict(fused_head_grads=True, fused_xproj=True)),
                    ("xproj only", dict(fused_lstm_grads=False, fused_xproj=True))):
    m = bench.make_model("dense", dev)
    for k, v in flags.items(): setattr(m, k, v)
    el, _ = bench.timed_steps(m, batch, 300, 20, 1, None, dev)
    print(f"{name:28s} {el / 300 * 1e3:.4f} ms/step  loss {m.train_step(batch).as_floats()['loss']:.4f}")

"""A/B of the step time under model attribute switches, arms interleaved in ONE process on one device (box-to-box and
run-to-run spread is ~2 %, so only same-process comparisons mean anything).
usage: ab_attr.py [dense|attention] name=value[,name=value] ...   (arm "base" = defaults is always included)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
wl = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("dense", "attention") else "dense"
arms = []
for a in sys.argv[1:]:
    if "=" in a:
        arms.append((a, {k: eval(v) for k, v in (kv.split("=") for kv in a.split(","))}))
arms.append(("base", {}))          # last: the first model of a process is not favoured by where its buffers land
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
models = []
for name, flags in arms:
    m = bench.make_model(wl, dev)
    for k, v in flags.items(): setattr(m, k, v)
    for _ in range(20): m.train_step(batch)
    models.append((name, m))
torch.cuda.synchronize()
res = {n: [] for n, _ in models}
for rnd in range(5):
    for name, m in models:
        el, _ = bench.timed_steps(m, batch, 200, 3, 1, None, dev)
        res[name].append(el / 200 * 1e3)
for name, v in res.items():
    v = sorted(v)
    print(f"{name:50s} median {v[len(v) // 2]:.4f}  min {v[0]:.4f}  max {v[-1]:.4f} ms/step")

"""Step time of config 2 under forced (tile, split-K) choices for the two products of one pair launch (ModelBase.g3_force),
all arms interleaved in ONE process like tools/ab_attr.py.  usage: pair_scan.py head|lstm
End of round 3: the cost model's pick (TN 128 x 80 + NT 64 x 128 split 4) is within 0.0002 ms of the best of the 28 pair forms
that exist; pair forms of the big tiles (TN 256 x 80 / 160 x 128 / 128 x 128 + NT 128 x 128, instantiated for the scan and
removed again) are 2.7 us SLOWER at their best ((8,1)+(3,4): one balanced round of 254 workgroups, one per CU) -- two small
workgroups per CU hide each other's stage boundaries better than one big one."""
import itertools, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "head"
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
if which == "head":      # kernel gradient TN 512 x 5001 x 960 (+ column sums: no split), input gradient NT 960 x 512 x 5001
    k1, k2 = (512, 5001, 960, True, False, 1), (960, 512, 5001, False, True, 1)
    c1 = [(4, 1), (5, 1), (7, 1)]
    c2 = [(t, s) for t in (5, 7) for s in (2, 3, 4, 5, 6, 8)]
else:                    # LSTM kernel + recurrent kernel gradients TN 512 x 2048 x 1024 x2, input gradient NT 1024 x 512 x 2048
    k1, k2 = (512, 2048, 1024, True, False, 2), (1024, 512, 2048, False, True, 1)
    c1 = [(5, 1), (7, 1)]
    c2 = [(t, s) for t in (5, 7) for s in (1, 2, 3, 4)]
arms = [("plan", None)] + [(f"{a}+{b}", {k1: a, k2: b}) for a, b in itertools.product(c1, c2)]
models = []
for name, force in arms:
    m = bench.make_model("dense", dev)
    if force: m.g3_force = force
    try:
        for _ in range(12): m.train_step(batch)
        torch.cuda.synchronize()
        models.append((name, m))
    except Exception as ex:           # a combination without a pair form falls back to two launches; one that cannot run is skipped
        print(f"{name}: skipped ({type(ex).__name__}: {str(ex)[:80]})")
res = {n: [] for n, _ in models}
for rnd in range(3):
    for name, m in models:
        el, _ = bench.timed_steps(m, batch, 150, 3, 1, None, dev)
        res[name].append(el / 150 * 1e3)
for name, v in sorted(res.items(), key=lambda kv: sorted(kv[1])[1]):
    v = sorted(v)
    print(f"{name:28s} median {v[1]:.4f}  min {v[0]:.4f} ms/step")

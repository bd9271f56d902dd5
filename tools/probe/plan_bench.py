import os, sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
for wl in ("dense", "attention"):
    for plan in (False, True, False, True):
        m = bench.make_model(wl, dev)
        m.plan_step = plan
        for _ in range(30): m.train_step(batch)
        torch.cuda.synchronize()
        res = []
        for r in range(3):
            el, per = bench.timed_steps(m, batch, 200, 3, 1, None, dev)
            res.append(el / 200 * 1e3)
        print(wl, "plan" if plan else "graph", " ".join(f"{v:.4f}" for v in res), flush=True)
        del m

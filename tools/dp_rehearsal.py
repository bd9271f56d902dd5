"""World-size-1 rehearsal of the data-parallel schedules over RCCL next to the single-GPU step, in one process:
what the schedule costs before any byte crosses a link.  usage: dp_rehearsal.py [dense|attention]"""
import os, sys
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from masters_thesis_amd import dp
wl = sys.argv[1] if len(sys.argv) > 1 else "dense"
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
batch, _ = bench.synth(0, dev)
single = bench.make_model(wl, dev)
arms = [("single-GPU step (one graph)", single)]
for name, env in (("DP schedule, ONE graph incl. collectives", "1"), ("DP schedule, segments (graphs / launch plans)", "0")):
    os.environ["TNT_DP_ONE_GRAPH"] = env
    m = bench.make_model(wl, dev)
    dp.attach(m, 1, rank=0)
    arms.append((name, m))
for _, m in arms:
    for _ in range(20): m.train_step(batch)
torch.cuda.synchronize()
res = {n: [] for n, _ in arms}
for rnd in range(4):
    for name, m in arms:
        el, _ = bench.timed_steps(m, batch, 200, 3, 1, None, dev)
        res[name].append(el / 200 * 1e3)
for (name, m) in arms:
    v = sorted(res[name])
    extra = ""
    if getattr(m, "grad_sync", None) is not None:
        extra = f"  one_graph={m.grad_sync.one_graph} err={m.grad_sync.capture_error}"
    print(f"{name:48s} median {v[len(v) // 2]:.4f} ms/step{extra}")
a, b = arms[0][1], arms[1][1]
print("max |theta(single) - theta(DP one graph)| =", (a.arena.theta - b.arena.theta).abs().max().item())
dist.destroy_process_group()

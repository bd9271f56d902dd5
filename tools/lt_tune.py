"""Produce the hipBLASLt solution table of csrc/blas.hip (g_lt_presets): run a few steps of both bench workloads with
TNT_LT_TUNE=1, under which tnt_gemm_lt_f32 times the heuristic's workspace-free candidates for every shape it meets and
prints the winner as a table line on stderr.  usage: TNT_LT_TUNE=1 python tools/lt_tune.py 2> table.txt"""
import os, sys
os.environ.setdefault("TNT_LT_TUNE", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
for wl in ("dense", "attention"):
    m = bench.make_model(wl, dev)
    if len(sys.argv) > 1:
        m.lt_min_flops = float(sys.argv[1])         # experiment: let smaller long-K products through as well
    for _ in range(4):
        m.train_step(batch)
    torch.cuda.synchronize()
    el, _ = bench.timed_steps(m, batch, 200, 3, 1, None, dev)
    print(f"{wl}: {el / 200 * 1e3:.4f} ms/step with the timed choices", flush=True)

"""Greedy-decode latency / throughput of both model forms at the BASELINE shapes (B = 64, max_len = 15)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
for wl in ("dense", "attention"):
    dev = torch.device("cuda", 0)
    model = bench.make_model(wl, dev, None)
    (data, tgt), _ = bench.synth(0, dev)
    x, cap, z, _ = data
    start = np.ones(bench.B, np.int64)
    for _ in range(3):
        model.greedy_predict(x, z, z, start, bench.T, bench.U, None)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        out = model.greedy_predict(x, z, z, start, bench.T, bench.U, None)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n
    print(f"{wl:9s}: {el * 1e3:7.3f} ms per batch of {bench.B} captions x {bench.T} tokens = {bench.B * bench.T / el:9.0f} tokens/s (incl. D2H of the outputs)")
    if wl == "attention":
        t0 = time.perf_counter()
        for _ in range(n):
            out = model.greedy_predict(x, z, z, start, bench.T, bench.U, None, return_s=False)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / n
        print(f"   return_s=False: {el * 1e3:7.3f} ms per batch = {bench.B * bench.T / el:9.0f} tokens/s")

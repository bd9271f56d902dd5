"""A/B timing of the config-2 step with model attributes toggled:  python tools/ab_bench.py fuse_tail=0 ..."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench

def run(**attrs):
    dev = torch.device("cuda", 0)
    model = bench.make_model("dense", dev, None)
    for k, v in attrs.items():
        setattr(model, k, v)
    batch, _ = bench.synth(0, dev)
    for _ in range(30):
        model.train_step(batch)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(200):
            model.train_step(batch)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200)
    return best * 1e3

if __name__ == "__main__":
    base = run()
    print(f"default            : {base:.4f} ms")
    for arg in sys.argv[1:]:
        k, v = arg.split("=")
        print(f"{arg:19s}: {run(**{k: eval(v)}):.4f} ms")

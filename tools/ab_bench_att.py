"""A/B timing of the config-3 (attention) step with model attributes toggled: python tools/ab_bench_att.py r_attn=0.0 ..."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench

def run(**attrs):
    dev = torch.device("cuda", 0)
    model = bench.make_model("attention", dev, None)
    for k, v in attrs.items():
        setattr(model, k, v)
    batch, _ = bench.synth(0, dev)
    for _ in range(30):
        model.train_step(batch)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(100):
            model.train_step(batch)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100)
    return best * 1e3

if __name__ == "__main__":
    print(f"default            : {run():.4f} ms")
    for arg in sys.argv[1:]:
        kv = dict(a.split("=") for a in arg.split(","))
        print(f"{arg:19s}: {run(**{k: eval(v) for k, v in kv.items()}):.4f} ms")

"""In-kernel phase timestamps of one step of the dense model's persistent LSTM forward (library built with -DTNT_LC_TRACE)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
m = bench.make_model("dense", dev)
for _ in range(30): m.train_step(batch)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
m.be.lib.tnt_debug_ls_trace.argtypes = [ctypes.c_void_p]
assert m.be.lib.tnt_debug_ls_trace(buf) == 0
t = list(buf)
names = {0: "top", 1: "h fragments in", 2: "MFMA done", 3: "partials in LDS, synced", 4: "reduced, gates", 5: "h out"}
base = t[0]
for k in names:
    print(f"{names[k]:26s} {(t[k] - base) * 10:7d} ns")

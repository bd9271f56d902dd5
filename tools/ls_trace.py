"""In-kernel phase timestamps of one step of the dense model's persistent LSTM chains (library built with -DTNT_LC_TRACE)."""
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
fwd = {0: "top", 1: "h fragments in", 2: "MFMA done", 3: "partials in LDS, synced", 4: "reduced, gates", 5: "h out"}
bwd = {8: "top", 9: "MFMA done, tiles pushed", 10: "32 partial tiles in", 11: "wave sums in LDS, synced", 12: "reduced, cell backward"}
for title, names in (("forward chain, step 5", fwd), ("BPTT chain, step 5", bwd)):
    print(title)
    base = t[min(names)]
    for k in names:
        print(f"  {names[k]:28s} {(t[k] - base) * 10:7d} ns")

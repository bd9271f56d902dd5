"""In-kernel phase timestamps of the persistent backward chain (build csrc with -DTNT_LC_TRACE: make CXXFLAGS+=... ;
wall_clock64 ticks of 10 ns).  One attention workgroup and one LSTM workgroup of row block 0, step T-3."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from masters_thesis_amd import _lib
dev = torch.device("cuda", 0)
batch, _ = bench.synth(0, dev)
m = bench.make_model("attention", dev)
for _ in range(30): m.train_step(batch)
torch.cuda.synchronize()
lib = _lib.load() if hasattr(_lib, "load") else m.be.lib
buf = (ctypes.c_ulonglong * 64)()
lib.tnt_debug_lc_trace.argtypes = [ctypes.c_void_p]
assert lib.tnt_debug_lc_trace(buf) == 0
t = list(buf)
bwd = {0: "A top", 7: "A poll starts", 1: "A parts in", 2: "A dctx", 3: "A dalpha + dot", 4: "A dot summed", 5: "A scores' + dq sums", 6: "A dq out",
       16: "L top", 17: "L pushed", 18: "L gathered", 19: "L dh in", 20: "L cell", 21: "L parts out", 25: "L step end", 23: "L resets issued", 24: "L resets drained"}
fwd = {32: "A top", 33: "A q parts in", 34: "A q", 36: "A sums", 37: "A ctx out",
       48: "L top", 49: "L h in", 50: "L h U done", 51: "L ctx in", 53: "L ctx W done", 54: "L synced", 55: "L gates",
       52: "L h out", 56: "L end barrier", 57: "L q part", 58: "L q out"}
print("backward attention: polls of the parts by thread 0 in that step:", t[8] + 1, "; the first ones return at (ns after the first):",
      [(x - t[26]) * 10 for x in t[26:26 + min(t[8] + 1, 6)]], "; first return", (t[26] - t[7]) * 10, "ns after the poll started,",
      (t[26] - t[21]) * 10, "ns after the traced LSTM workgroup's publish")
for title, names in (("forward chain, step 5", fwd), ("backward chain, step 5", bwd)):
    base = min(t[k] for k in names if t[k])
    print(title)
    for k in sorted(names, key=lambda k: t[k]):
        print(f"  {names[k]:14s} {(t[k] - base) * 10:8d} ns")

# per-step timeline (tnt_debug_lc_steps): entry, behind the launch's flag barrier, every step's mark, loop done, outputs stored
sb = (ctypes.c_ulonglong * 160)()
lib.tnt_debug_lc_steps.argtypes = [ctypes.c_void_p]
assert lib.tnt_debug_lc_steps(sb) == 0
T = 15          # bench.synth: caption length 16 -> 15 chain steps
for role, name in enumerate(("forward attention (mark: h in)", "forward LSTM (mark: h out)", "backward attention (mark: dq out)",
                             "backward LSTM (mark: parts out)")):
    s = list(sb)[role * 40:(role + 1) * 40]
    marks = s[2:2 + T]
    print(f"{name}: entry -> barrier {(s[1] - s[0]) * 10} ns, -> first mark {(marks[0] - s[0]) * 10} ns; step periods (ns): "
          + " ".join(str((b - a) * 10) for a, b in zip(marks, marks[1:]))
          + f"; last mark -> loop done {(s[38] - marks[-1]) * 10} ns -> stored {(s[39] - s[38]) * 10} ns; entry -> stored {(s[39] - s[0]) * 10} ns")

pb = (ctypes.c_ulonglong * 320)()
lib.tnt_debug_lc_pub.argtypes = [ctypes.c_void_p]
assert lib.tnt_debug_lc_pub(pb) == 0
pb = list(pb)
base = min(x for x in pb if x > 1000)
print("backward step 5, row block 0, ns after the earliest mark:")
print("  LSTM workgroups: step top     ", [(x - base) * 10 for x in pb[96:112]])
print("  LSTM workgroups: pushed       ", [(x - base) * 10 for x in pb[128:144]])
print("  gather loop left by wave 0    ", [(x - base) * 10 for x in pb[192:208]])
print("  gather loop left by wave 4    ", [(x - base) * 10 for x in pb[224:240]])
print("  gather loop left by wave 8    ", [(x - base) * 10 for x in pb[256:272]])
print("  gather rounds of wave 4       ", pb[288:304])
print("  LSTM workgroups: gathered     ", [(x - base) * 10 for x in pb[160:176]])
print("  LSTM workgroups: dh_att in    ", [(x - base) * 10 for x in pb[0:16]])
print("  LSTM workgroups: parts out    ", [(x - base) * 10 for x in pb[32:48]])
print("  attention workgroups: dq out  ", [(x - base) * 10 for x in pb[64:72]])

"""Per-dispatch view of ONE steady-state step from a rocprofv3 --kernel-trace CSV:
prints the kernels of the last complete step in launch order with start offset and duration."""
import csv, glob, sys
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "step_tick_kernel"
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-3] + 1, idx[-2] + 1          # one full step between two markers
t0 = int(rows[a]["Start_Timestamp"])
tot = 0
prev_end = t0
print(f"# {f}: dispatches {a}..{b}")
print(f"{'start_us':>9s} {'gap_us':>7s} {'dur_us':>8s}  kernel (grid)")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} {(s - prev_end) / 1e3:7.1f} {(e - s) / 1e3:8.2f}  {n} ({r.get('Grid_Size_X', '?')})")
    tot += e - s
    prev_end = e
print(f"# span {(prev_end - t0) / 1e3:.1f} us, kernel time {tot / 1e3:.1f} us")

"""Summarise a rocprofv3 --kernel-trace --stats CSV into a short per-kernel table (per step)."""
import csv, glob, sys
d, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# source: {f}\n# steps profiled: {steps}; total kernel time per step: {tot / steps / 1e3:.1f} us")
print(f"{'kernel':100s} {'calls/step':>10s} {'avg_us':>9s} {'us/step':>9s} {'%':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    n = r["Name"].replace("(anonymous namespace)::", "")[:100]
    print(f"{n:100s} {int(r['Calls']) / steps:10.2f} {float(r['AverageNs']) / 1e3:9.2f} "
          f"{float(r['TotalDurationNs']) / steps / 1e3:9.1f} {100 * float(r['TotalDurationNs']) / tot:6.1f}")

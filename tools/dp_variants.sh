for v in "A,B2,C0,C1,C2" "B2,C0,C1,C2" "C0,C1,C2" "none"; do
  TNT_DP_EAGER=$v timeout -k 10 200 python bench.py --force-dp --no-config3 --steps 300 --warmup 30 > gpurun_out/b_dpv.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/b_dpv.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$v", d["ms_per_step"], d["config"]["step_ms_p10_p50_p90"])
PY
done

# plan replay vs hipGraph replay of the training step with the host's cores busy (16 spinning processes): how much of the
# launch plan's advantage survives a loaded host.  usage (inside gpurun): bash tools/probe/plan_bench_loaded.sh
for i in $(seq 1 16); do timeout -k 5 150 python3 -c "
while True: pass" & done
sleep 1
python3 tools/probe/plan_bench.py 2>&1 | tail -8
wait

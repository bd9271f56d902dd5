# per-dispatch trace of one steady-state step of the world-size-1 DP rehearsal (rocprofv3 --kernel-trace)
# usage (inside gpurun): bash tools/dp_trace_quick.sh dense|attention
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/traceq
rm -rf $O/dp_$1; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/dp_$1 -- python3 $R/tools/dp_trace_run.py $1 > $O/dp_$1.log 2>&1
python3 $R/tools/trace_step.py $O/dp_$1 ${2:-adam_fin_kernel} > $O/dp_$1_step.txt
rm -rf $O/dp_$1
cat $O/dp_$1_step.txt

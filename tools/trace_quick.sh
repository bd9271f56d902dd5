# per-dispatch trace of one steady-state step (rocprofv3 --kernel-trace): usage (inside gpurun): bash tools/trace_quick.sh dense|attention [marker]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/traceq
rm -rf $O/$1; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/$1 -- python3 $R/tools/prof_step.py $1 100 > $O/$1.log 2>&1
python3 $R/tools/trace_step.py $O/$1 ${2:-adam_fin_kernel} > $O/$1_step.txt
rm -rf $O/$1
cat $O/$1_step.txt

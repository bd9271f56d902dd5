# Round-3 profile set, one gpurun call: rocprofv3 kernel stats of 400 training steps of each bench workload, the PMC passes
# (tools/pmc_round3.sh), and the in-kernel phase trace of the attention model's chains (trace library built out of tree by
# tools/lc_trace_build.sh before the call).  Summaries land in gpurun_out/prof3/; copy what is to be judged to profiles/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof3
rm -rf $O; mkdir -p $O
for wl in dense attention; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$wl -- python3 $R/tools/prof_step.py $wl 400 > $O/$wl.log 2>&1
  python3 $R/tools/prof_summary.py $O/$wl 400 > $O/${wl}_summary.txt
done
cd $R
if [ -f .trace_build/pkg/csrc/libtnt_hip.so ]; then
  TNT_HIP_LIB=$R/.trace_build/pkg/csrc/libtnt_hip.so python3 tools/lc_trace.py > $O/lc_trace.txt 2>&1 || true
fi
sh tools/pmc_round3.sh > $O/pmc.log 2>&1
cp gpurun_out/pmc3/summary.txt $O/pmc_summary.txt
head -24 $O/dense_summary.txt; head -30 $O/attention_summary.txt

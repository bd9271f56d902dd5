# rocprofv3 kernel stats of 400 training steps of one or both bench workloads (no PMC passes): gpurun_out/profq/<wl>_summary.txt
# usage (inside gpurun): sh tools/prof_quick.sh [dense|attention ...]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profq
mkdir -p $O
for wl in ${@:-dense attention}; do
  rm -rf $O/$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$wl -- python3 $R/tools/prof_step.py $wl 400 > $O/$wl.log 2>&1
  python3 $R/tools/prof_summary.py $O/$wl 400 > $O/${wl}_summary.txt
  rm -rf $O/$wl
  head -28 $O/${wl}_summary.txt | cut -c1-150
done

set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof2
rm -rf $O; mkdir -p $O
for wl in dense attention; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$wl -- python3 $R/tools/prof_step.py $wl 400 > $O/$wl.log 2>&1
  python3 $R/tools/prof_summary.py $O/$wl 400 > $O/${wl}_summary.txt
done
head -30 $O/dense_summary.txt

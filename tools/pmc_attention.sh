# HBM traffic of the attention model's chain kernels (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_att
rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/a -- python3 $R/tools/prof_step.py attention 12 > $O/a.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python3 $R/tools/prof_step.py attention 12 > $O/b.log 2>&1
cd $R
for k in lc_seq_fwd_kernel lc_seq_bwd_kernel; do echo "== $k"; for p in a b; do python3 tools/pmc_summary.py gpurun_out/pmc_att/$p $k; done; done > gpurun_out/pmc_att/summary.txt
cat gpurun_out/pmc_att/summary.txt

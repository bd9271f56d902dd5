# rocprofv3 --pmc passes (separate runs: FETCH_SIZE | WRITE_SIZE GRBM_GUI_ACTIVE | SQ busy counters) over 12 training steps of
# each bench workload: HBM traffic and MFMA-busy of the gemm3 kernels and the chains inside the step (round 3).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc3
rm -rf $O; mkdir -p $O
for wl in dense attention; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${wl}_a -- python3 $R/tools/prof_step.py $wl 12 > $O/${wl}_a.log 2>&1
  rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/${wl}_b -- python3 $R/tools/prof_step.py $wl 12 > $O/${wl}_b.log 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/${wl}_c -- python3 $R/tools/prof_step.py $wl 12 > $O/${wl}_c.log 2>&1
done
cd $R
for wl in dense attention; do
  for k in gemm3_pair_kernel gemm3_kernel lstm_seq_fwd_kernel lstm_seq_bwd_kernel lc_seq_fwd_kernel lc_seq_bwd_kernel dense_dw_skinny_kernel adam_fin_kernel; do
    echo "== $wl $k"; for p in a b c; do python3 tools/pmc_summary.py gpurun_out/pmc3/${wl}_$p $k; done
  done
done > gpurun_out/pmc3/summary.txt
cat gpurun_out/pmc3/summary.txt

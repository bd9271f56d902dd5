set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
rm -rf $O; mkdir -p $O
for s in lstm_seq_one enc_one; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${s}_a -- python3 $R/tools/$s.py > $O/${s}_a.log 2>&1
  rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/${s}_b -- python3 $R/tools/$s.py > $O/${s}_b.log 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/${s}_c -- python3 $R/tools/$s.py > $O/${s}_c.log 2>&1
done
cd $R
for k in lstm_seq_fwd_kernel lstm_seq_bwd_kernel; do echo "== $k"; for p in a b c; do python3 tools/pmc_summary.py gpurun_out/pmc/lstm_seq_one_$p $k; done; done > gpurun_out/pmc/summary.txt
for k in dense_fwd_stream_kernel dense_gram_norm_kernel "dense_dw_skinny_kernelILi4ELb1ELi2" "dense_dw_skinny_kernelILi4ELb1ELi0"; do echo "== $k"; for p in a b c; do python3 tools/pmc_summary.py gpurun_out/pmc/enc_one_$p $k; done; done >> gpurun_out/pmc/summary.txt
cat gpurun_out/pmc/summary.txt

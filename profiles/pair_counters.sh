# Counters of the float32 pair kernels at two sides (2^20 pairs per call, profiles/pair_one_side.py):
# one rocprofv3 --pmc pass per set, nothing else traced.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for side in 12 31; do
  for set in "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA"; do
    tag=${side}_$(echo $set | cut -c1-12 | tr ' ' '_')
    timeout -s KILL 120 rocprofv3 --pmc $set -d $R/gpurun_out/pairpmc/$tag -o p -- python3 $R/profiles/pair_one_side.py $side > $R/gpurun_out/pairpmc_$tag.log 2>&1 || { echo "failed: side $side set $set"; tail -3 $R/gpurun_out/pairpmc_$tag.log; continue; }
    echo "== side $side"
    python3 $R/profiles/pmc_summary.py $(find $R/gpurun_out/pairpmc/$tag -name "*.db" | head -1) scan_pairs
  done
done

#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) on the fc1 grouped GEMM the DEFAULT path launches -> gpurun_out/pmc_fc1_<counter>.csv (gemm rows only)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python "$GRAFT_REPO_ROOT/tools/gemm_pmc_target.py" ) > gpurun_out/pmc_fc1_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
  grep -E 'Counter_Name|gemm' "$f" | cut -c1-700 > gpurun_out/pmc_fc1_$tag.csv
done

#!/bin/bash
# PMC passes (separate runs, --kernel-trace only -- never combined with other trace domains) -> gpurun_out/<tag>_pmc_<target>_<counter>.csv
# usage: tools/gpu_pmc.sh <tag> fc1 [attn_bwd vit_fwd]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
rtag=$1; shift
for t in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo $c | cut -d' ' -f1)
    rm -rf /tmp/pmc_${t}_$tag
    ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${t}_$tag -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_targets.py" $t ) > gpurun_out/${rtag}_pmc_${t}_$tag.log 2>&1
    f=$(find /tmp/pmc_${t}_$tag -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && grep -E 'Counter_Name|gemm3|attn_' "$f" | cut -c1-900 > gpurun_out/${rtag}_pmc_${t}_$tag.csv
  done
done
ls -la gpurun_out/${rtag}_pmc_*.csv | head -20

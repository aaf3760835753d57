#!/bin/bash
# round 4, session 3: attn_fwd3 on hardware -- parity cases, A/B against attn_fwd2, LDS bank conflicts of the new tile images
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -8 ) > gpurun_out/r04_s3_pytest_attn.log 2>&1
timeout 600 python tools/probes/attn_fwd3_ab.py > gpurun_out/r04_attn_fwd3_ab.json 2> gpurun_out/r04_attn_fwd3_ab.err
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pmc_v
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_v -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_targets.py" vit_fwd ) > gpurun_out/r04_s3_pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_v -name '*counter_collection.csv' | head -1); k=$(find /tmp/pmc_v -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && grep -E 'Counter_Name|attn_' "$f" | cut -c1-900 > gpurun_out/r04_s3_pmc_vit_fwd3_$tag.csv
  [ -n "$k" ] && grep -E 'Kernel_Name|attn_' "$k" | cut -c1-900 > gpurun_out/r04_s3_trace_vit_fwd3_$tag.csv
done
tail -4 gpurun_out/r04_s3_pytest_attn.log; cat gpurun_out/r04_attn_fwd3_ab.json; tail -3 gpurun_out/r04_attn_fwd3_ab.err

#!/bin/bash
# round 4, session 2: attention-forward ablation (where a key tile's time goes), grouped tile orders (default / ragged-last / ragged-first:
# TF/s and fabric traffic), average shader clock of the three roofline kernels (GRBM_GUI_ACTIVE / duration)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/probes/attn_fwd_ablate.py > gpurun_out/r04_attn_fwd_ablate.json 2> gpurun_out/r04_attn_fwd_ablate.err
timeout 600 python tools/probes/ragged_last_ab.py > gpurun_out/r04_order_ab.json 2> gpurun_out/r04_order_ab.err
for ord in 516 2564; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    rm -rf /tmp/pmc_o
    ( cd /tmp && ARIA_GEMM_ORDER=$ord timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_o -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_targets.py" fc1 ) > gpurun_out/r04_pmc_order${ord}_$tag.log 2>&1
    f=$(find /tmp/pmc_o -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && grep -E 'Counter_Name|gemm3' "$f" | cut -c1-900 > gpurun_out/r04_pmc_order${ord}_$tag.csv
  done
done
for t in fc1 vit_fwd attn_bwd; do
  rm -rf /tmp/pmc_c
  ( cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_c -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_targets.py" $t ) > gpurun_out/r04_clock_$t.log 2>&1
  f=$(find /tmp/pmc_c -name '*counter_collection.csv' | head -1); k=$(find /tmp/pmc_c -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && grep -E 'Counter_Name|gemm3|attn_' "$f" | cut -c1-900 > gpurun_out/r04_clock_${t}_counter.csv
  [ -n "$k" ] && grep -E 'Kernel_Name|gemm3|attn_' "$k" | cut -c1-900 > gpurun_out/r04_clock_${t}_trace.csv
done
cut -c1-1500 gpurun_out/r04_attn_fwd_ablate.json; tail -2 gpurun_out/r04_attn_fwd_ablate.err; cat gpurun_out/r04_order_ab.json; tail -2 gpurun_out/r04_order_ab.err

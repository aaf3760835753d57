#!/bin/bash
# Round-2 GPU session 1: hardware parity (incl. the new full-width cases), baseline bench line, the checks round 1 left for hardware,
# and the PMC passes of the SHIPPED fc1 kernel (tile order as shipped).  Everything lands in gpurun_out/.
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/s1_host.txt; free -g >> gpurun_out/s1_host.txt
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 ) > gpurun_out/s1_pytest.log 2>&1
( time timeout 600 python bench.py --steps 5 --warmup 2 --time-grouped ) > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err
timeout 600 python tools/next_round_gpu_checks.py > gpurun_out/next_round_checks.json 2> gpurun_out/s1_checks.err
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/s1_pmc_$c" -o p -- python "$GRAFT_REPO_ROOT/tools/gemm_pmc_target.py" ) > gpurun_out/s1_pmc_$c.log 2>&1
done
( cd /tmp && timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/s1_pmc_TCC" -o p -- python "$GRAFT_REPO_ROOT/tools/gemm_pmc_target.py" ) > gpurun_out/s1_pmc_TCC.log 2>&1
# keep only the grouped-GEMM rows of the counter CSVs (the files are huge otherwise)
for d in gpurun_out/s1_pmc_*/; do
  for f in $(find $d -name '*counter_collection.csv'); do grep -E 'Counter_Name|gemm' "$f" | cut -c1-600 > "$f.gemm.csv"; rm -f "$f"; done
  find $d -name '*kernel_trace.csv' -delete; find $d -name '*.db' -delete
done
ls -la gpurun_out | tail -30

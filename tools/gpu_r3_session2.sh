#!/bin/bash
# Round-3 GPU session 2: hardware parity of the re-staged attention backward kernels (LDS-DMA tiles, dQ v5) and their same-box A/B against
# the round-2 library, per-kernel times at 64K (rocprofv3), then the driver's bench line.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullwidth.py -m gpu -q -k "attention" 2>&1 | tail -15 ) > gpurun_out/s2_pytest_attn.log 2>&1
timeout 600 python tools/probes/attn_bwd_r3_ab.py > gpurun_out/s2_attn_ab.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/s2_prof_attn" -o p -- python "$GRAFT_REPO_ROOT/tools/probes/attn_bwd_r3_ab.py" ) > gpurun_out/s2_prof_attn.log 2>&1
db=$(find gpurun_out/s2_prof_attn -name '*.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" 12 > gpurun_out/s2_attn_kernel_stats.txt 2>&1; rm -rf gpurun_out/s2_prof_attn
( time timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench.err
tail -4 gpurun_out/s2_pytest_attn.log; cat gpurun_out/s2_attn_ab.log | tail -5

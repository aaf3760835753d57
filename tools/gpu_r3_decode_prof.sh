#!/bin/bash
# rocprofv3 kernel trace of the decode engine's model step (6-launch schedule, 53 tokens at position 280..): per-kernel durations
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_dec" -o p -- python "$GRAFT_REPO_ROOT/tools/decode_bench.py" --only=engine_fused6 ) > gpurun_out/prof_dec.log 2>&1
db=$(find gpurun_out/prof_dec -name '*.db' | head -1)
python tools/rocpd_stats.py "$db" 30 > gpurun_out/kernel_stats_decode_r03.txt 2>&1
rm -rf gpurun_out/prof_dec; grep "^{" gpurun_out/prof_dec.log; head -16 gpurun_out/kernel_stats_decode_r03.txt | cut -c1-150

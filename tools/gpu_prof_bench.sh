#!/bin/bash
# rocprofv3 kernel trace of the default bench.py command -> per-kernel summary (tools/rocpd_stats.py).  $1 = tag, rest = extra bench args
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; tag=$1; shift
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" ) > gpurun_out/prof_$tag.log 2>&1
db=$(find gpurun_out/prof_$tag -name '*.db' | head -1)
python tools/rocpd_stats.py "$db" 60 gemm2_kernel gemm3_kernel > gpurun_out/kernel_stats_$tag.txt 2>&1
grep '^{' gpurun_out/prof_$tag.log > gpurun_out/bench_under_rocprof_$tag.json
rm -rf gpurun_out/prof_$tag

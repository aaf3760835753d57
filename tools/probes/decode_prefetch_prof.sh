#!/bin/bash
# per-kernel durations of the decode step with the graph's prefetch branch off / on (rocprofv3 --kernel-trace --stats, one run per arm)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=$1; shift
for arm in "$@"; do
  rm -rf /tmp/dp_$arm
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/dp_$arm -o p -- python "$GRAFT_REPO_ROOT/tools/probes/decode_prefetch_ab.py" $arm ) > gpurun_out/${tag}_dp_$arm.log 2>&1
  db=$(find /tmp/dp_$arm -name '*.db' | head -1)
  python tools/rocpd_stats.py "$db" 14 > gpurun_out/${tag}_dp_stats_$arm.txt 2>&1
  head -16 gpurun_out/${tag}_dp_stats_$arm.txt | cut -c1-50,112-175
done

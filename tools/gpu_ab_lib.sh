#!/bin/bash
# Same-box A/B of two builds of the library (boxes of the pool differ by up to 10 %): "$@" (default: bench.py) with the tree's library,
# then with build/old/libaria_hip.so copied over it (the box's tree is a scratch copy), then the tree's again.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
cp aria_amd/libaria_hip.so /tmp/new.so
if [ $# -eq 0 ]; then
  run() { python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['achieved'])"; }
else
  CMD="$*"
  run() { echo "$1 $($CMD 2>/dev/null | tail -1)"; }
fi
run new1 | tee gpurun_out/ab.log
cp build/old/libaria_hip.so aria_amd/libaria_hip.so; run old | tee -a gpurun_out/ab.log
cp /tmp/new.so aria_amd/libaria_hip.so; run new2 | tee -a gpurun_out/ab.log

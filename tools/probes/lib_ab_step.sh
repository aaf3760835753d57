#!/bin/bash
# Same-box interleaved A/B of the config #3 step between the tree's library and another build of it (boxes of the pool differ by several %):
#   gpurun -- 'bash tools/probes/lib_ab_step.sh <tag> <other.so> [pairs]'  ->  gpurun_out/<tag>_lib_ab.json
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1; other=$2; pairs=${3:-2}
cp aria_amd/libaria_hip.so /tmp/tree.so
run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-long64k --no-inference-records --no-fusions-ab --no-launch-classes --no-lora-record 2>/dev/null | tail -1 \
        | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'arm': '$1', 'ms_per_step': d['ms_per_step'], 'fc1_swiglu_TFs': d['roofline']['achieved']}))"; }
: > gpurun_out/${tag}_lib_ab.jsonl
for i in $(seq 1 $pairs); do
  cp "$other" aria_amd/libaria_hip.so; run other | tee -a gpurun_out/${tag}_lib_ab.jsonl
  cp /tmp/tree.so aria_amd/libaria_hip.so; run tree | tee -a gpurun_out/${tag}_lib_ab.jsonl
done
cp /tmp/tree.so aria_amd/libaria_hip.so

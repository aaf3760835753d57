#!/bin/bash
# Same-box interleaved A/B of the config #3 step between two environments (e.g. tile-order words): the step is power-limited, so a launch-level
# result need not carry over -- this is the test that counts.
#   gpurun -- 'bash tools/probes/env_ab_step.sh <tag> "<VAR=VALUE ...>" [pairs]'   ->  gpurun_out/<tag>_env_ab.jsonl   (arm A = unset)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1; envb=$2; pairs=${3:-2}
run() { env $2 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-long64k --no-inference-records --no-fusions-ab --no-launch-classes --no-lora-record 2>/dev/null | tail -1 \
        | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'arm': '$1', 'ms_per_step': d['ms_per_step'], 'fc1_swiglu_TFs': d['roofline']['achieved']}))"; }
: > gpurun_out/${tag}_env_ab.jsonl
for i in $(seq 1 $pairs); do
  run "default" "" | tee -a gpurun_out/${tag}_env_ab.jsonl
  run "$envb" "$envb" | tee -a gpurun_out/${tag}_env_ab.jsonl
done

#!/bin/bash
# round 4, session 34: the default bench.py line on the final tree (roofline.traffic now from the PMC pass of HEAD)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 400 python bench.py > gpurun_out/r04_bench_final2.json 2> gpurun_out/r04_bench_final2.err ) 2> gpurun_out/r04_bench_final2.time
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_final2.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["recipe_grad_checkpointing"]["ms_per_step"], d["long64k"]["ms_per_step"],
      d["generate_config2"]["value"], d["generate_config2"]["decode_ms_per_token"], d["prefill_config4"]["value"], d["cpu_baseline"]["value"])
PY
tail -3 gpurun_out/r04_bench_final2.time

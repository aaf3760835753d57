#!/bin/bash
# round 4, session 16: cross-XCD visibility of stores inside a launch, re-read rates (infinity cache) -- probes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python tools/probes/xcd_visibility.py > gpurun_out/r04_xcd_visibility.json 2> gpurun_out/r04_xcd_visibility.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_xcd_visibility.json"))
for k, v in d["visibility"].items():
    print(v["stale_words_of_128000"], v["timed_out"], k)
for r in d["reread"]:
    print(r)
PY
tail -3 gpurun_out/r04_xcd_visibility.err

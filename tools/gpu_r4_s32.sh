#!/bin/bash
# round 4, session 32: HEAD's streamed decode schedule once more on the 28-layer model (after the clean-up commits): equality + time, plain and captured
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 200 python tools/probes/decode_stream_ab.py --graph > gpurun_out/r04_decode_stream_ab_head.json 2> gpurun_out/r04_decode_stream_ab_head.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_decode_stream_ab_head.json"))
print({k: (v["ms_per_token"], v["logits_equal_first_run"], v["error_word"]) for k, v in d["runs"].items()})
PY
tail -2 gpurun_out/r04_decode_stream_ab_head.err

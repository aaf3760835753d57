#!/bin/bash
# round 4, session 22: streamed decode with the attention states on the write-back path (ST mask 63): equality + time; the stage timeline of a token
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python tools/probes/decode_stream_ab.py > gpurun_out/r04_decode_stream_ab6.json 2> gpurun_out/r04_decode_stream_ab6.err
timeout 300 python tools/probes/decode_stream_timeline.py --lib=build/abl/libaria_decode_tl.so > gpurun_out/r04_decode_stream_timeline.json 2> gpurun_out/r04_decode_stream_timeline.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_decode_stream_ab6.json"))
print({k: (v["ms_per_token"], v["logits_equal_first_run"], v["error_word"], v["max_abs_diff_vs_first_run"]) for k, v in d["runs"].items()})
PY
cat gpurun_out/r04_decode_stream_timeline.json; tail -3 gpurun_out/r04_decode_stream_timeline.err

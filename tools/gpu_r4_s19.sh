#!/bin/bash
# round 4, session 19: streamed decode with RETURNING exchanges as the stores (is the 28-layer mismatch a fire-and-forget acknowledgement that
# arrives before the other XCDs can see the data?)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 420 python tools/probes/decode_stream_ab.py > gpurun_out/r04_decode_stream_ab5.json 2> gpurun_out/r04_decode_stream_ab5.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_decode_stream_ab5.json"))
print({k: (v["ms_per_token"], v["logits_equal_first_run"], v["error_word"], v["max_abs_diff_vs_first_run"]) for k, v in d["runs"].items()})
PY
tail -2 gpurun_out/r04_decode_stream_ab5.err

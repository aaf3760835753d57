#!/bin/bash
# round 4, session 20: which stage of the streamed decode schedule needs an L2 write-back for the 28-layer token to equal the launch schedule
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for m in 1 2 4 8 16 32 64; do
  timeout 200 python tools/probes/decode_stream_ab.py --lib=build/abl/libaria_decode_wb$m.so > gpurun_out/r04_decode_stream_wb$m.json 2> gpurun_out/r04_decode_stream_wb$m.err
done
python - <<'PY'
import json
for m in (1, 2, 4, 8, 16, 32, 64):
    try:
        d = json.load(open(f"gpurun_out/r04_decode_stream_wb{m}.json"))
        v = d["runs"]["streamed #1"]
        print(m, v["ms_per_token"], v["logits_equal_first_run"], v["error_word"], v["max_abs_diff_vs_first_run"])
    except Exception as e:
        print(m, "unreadable", e)
PY

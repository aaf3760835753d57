#!/bin/bash
# round 4, session 27: what the streamed decode kernel's workgroups spend their time on -- timing-only builds without waits and counters (3), and
# additionally without the activation vector (8), the output exchanges (16), the down-projection's activation images (32)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in a3 a11 a19 a35 a59; do
timeout 200 python tools/probes/decode_stream_ab.py --lib=build/abl/libaria_decode_$v.so > gpurun_out/r04_decode_stream_abl_$v.json 2> gpurun_out/r04_decode_stream_abl_$v.err
done
python - <<'PY'
import json
for v in ("a3", "a11", "a19", "a35", "a59"):
    d = json.load(open(f"gpurun_out/r04_decode_stream_abl_{v}.json"))
    print(v, d["runs"]["streamed #1"]["ms_per_token"], d["runs"]["launch6 #1"]["ms_per_token"])
PY

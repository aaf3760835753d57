#!/bin/bash
# round 4, session 28: mean time per phase of a streamed decode workgroup (timeline builds with and without the waits)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in tl tlnosync; do
timeout 120 python tools/probes/decode_stream_timeline.py --lib=build/abl/libaria_decode_$v.so > gpurun_out/r04_decode_stream_phases_$v.json 2> gpurun_out/r04_decode_stream_phases_$v.err
python - <<PY
import json
d = json.load(open("gpurun_out/r04_decode_stream_phases_$v.json"))
print("$v", "per layer us", d["per_layer_us"], "error", d["error_word"])
for k, v in d["mean_workgroup_phases_us"].items():
    print(" ", k, v)
PY
done

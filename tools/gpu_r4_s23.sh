#!/bin/bash
# round 4, session 23: streamed decode, completion through 64 flag lines per stage (nobody polls a counter): equality + time, toy tests, no-sync floor, timeline
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -k "streamed" 2>&1 | tail -4 ) > gpurun_out/r04_s23_pytest.log 2>&1
timeout 300 python tools/probes/decode_stream_ab.py --graph > gpurun_out/r04_decode_stream_ab7.json 2> gpurun_out/r04_decode_stream_ab7.err
timeout 300 python tools/probes/decode_stream_ab.py --lib=build/abl/libaria_decode_nosync.so > gpurun_out/r04_decode_stream_ab7_nosync.json 2> gpurun_out/r04_decode_stream_ab7_nosync.err
timeout 300 python tools/probes/decode_stream_timeline.py --lib=build/abl/libaria_decode_tl.so > gpurun_out/r04_decode_stream_timeline2.json 2> gpurun_out/r04_decode_stream_timeline2.err
tail -2 gpurun_out/r04_s23_pytest.log
python - <<'PY'
import json
for f in ("gpurun_out/r04_decode_stream_ab7.json", "gpurun_out/r04_decode_stream_ab7_nosync.json"):
    d = json.load(open(f))
    print({k: (v["ms_per_token"], v["logits_equal_first_run"], v["error_word"], v["max_abs_diff_vs_first_run"]) for k, v in d["runs"].items()})
d = json.load(open("gpurun_out/r04_decode_stream_timeline2.json"))
print("per layer us", d["per_layer_us"], "error", d["error_word"])
for layer in ("2", "13"):
    print(layer, {k: (v["first_resident"], v["last_done"]) for k, v in d["layers"][layer].items()})
PY
tail -2 gpurun_out/r04_decode_stream_timeline2.err

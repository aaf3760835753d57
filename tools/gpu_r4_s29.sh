#!/bin/bash
# round 4, session 29: streamed decode, workgroups that end without waiting (self-validating data in per-layer buffers, unordered completion hints,
# lead workgroups raising the flags): equality + time, phases
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 200 python -m pytest tests/test_gpu_model.py -m gpu -q -k "streamed" 2>&1 | tail -4 ) > gpurun_out/r04_s29_pytest.log 2>&1
timeout 200 python tools/probes/decode_stream_ab.py --graph > gpurun_out/r04_decode_stream_ab11.json 2> gpurun_out/r04_decode_stream_ab11.err
timeout 120 python tools/probes/decode_stream_timeline.py --lib=build/abl/libaria_decode_tl.so > gpurun_out/r04_decode_stream_phases2.json 2> gpurun_out/r04_decode_stream_phases2.err
tail -2 gpurun_out/r04_s29_pytest.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_decode_stream_ab11.json"))
print({k: (v["ms_per_token"], v["logits_equal_first_run"], v["error_word"], v["max_abs_diff_vs_first_run"]) for k, v in d["runs"].items()})
d = json.load(open("gpurun_out/r04_decode_stream_phases2.json"))
print("per layer us", d["per_layer_us"], "error", d["error_word"])
for k, v in d["mean_workgroup_phases_us"].items():
    print(" ", k, v)
PY
tail -2 gpurun_out/r04_decode_stream_ab11.err

#!/bin/bash
# round 4, session 24: streamed decode, iteration 3 (counter tree, one row pair per wave in the up-projections, two-phase down-projection rows,
# four waves per SIMD): equality + time, three-waves variant, no-sync floor, timeline
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -k "streamed" 2>&1 | tail -4 ) > gpurun_out/r04_s24_pytest.log 2>&1
timeout 300 python tools/probes/decode_stream_ab.py --graph > gpurun_out/r04_decode_stream_ab8.json 2> gpurun_out/r04_decode_stream_ab8.err
for v in occ3 nosync; do
timeout 300 python tools/probes/decode_stream_ab.py --lib=build/abl/libaria_decode_$v.so > gpurun_out/r04_decode_stream_ab8_$v.json 2> gpurun_out/r04_decode_stream_ab8_$v.err
done
timeout 300 python tools/probes/decode_stream_timeline.py --lib=build/abl/libaria_decode_tl.so > gpurun_out/r04_decode_stream_timeline3.json 2> gpurun_out/r04_decode_stream_timeline3.err
tail -2 gpurun_out/r04_s24_pytest.log
python - <<'PY'
import json
for f in ("gpurun_out/r04_decode_stream_ab8.json", "gpurun_out/r04_decode_stream_ab8_occ3.json", "gpurun_out/r04_decode_stream_ab8_nosync.json"):
    d = json.load(open(f))
    print(f, {k: (v["ms_per_token"], v["logits_equal_first_run"], v["error_word"], v["max_abs_diff_vs_first_run"]) for k, v in d["runs"].items()})
d = json.load(open("gpurun_out/r04_decode_stream_timeline3.json"))
print("per layer us", d["per_layer_us"], "error", d["error_word"])
for layer in ("2", "13"):
    print(layer, {k: (v["first_resident"], v["last_done"]) for k, v in d["layers"][layer].items()})
PY
tail -2 gpurun_out/r04_decode_stream_timeline3.err

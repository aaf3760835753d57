#!/bin/bash
# round 4, session 17: the lean streamed decode schedule (index order, fire-and-forget completion, two rows per wave / three waves per SIMD):
# bit-equality on hardware, A/B on the 25.3 B model; which half of the visibility protocol the 28-layer mismatch of session 14 lives in
# (loads / stores / LDS-DMA variants), the ticket's cost, the no-sync floor
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 420 python -m pytest tests/test_gpu_model.py -m gpu -q -k "streamed" 2>&1 | tail -12 ) > gpurun_out/r04_s18_pytest.log 2>&1
timeout 420 python tools/probes/decode_stream_ab.py > gpurun_out/r04_decode_stream_ab4.json 2> gpurun_out/r04_decode_stream_ab4.err
for v in far32 near2 st0 nosync; do
  timeout 300 python tools/probes/decode_stream_ab.py --lib=build/abl/libaria_decode_$v.so > gpurun_out/r04_decode_stream_ab4_$v.json 2> gpurun_out/r04_decode_stream_ab4_$v.err
done
tail -5 gpurun_out/r04_s18_pytest.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_decode_stream_ab4*.json")):
    try:
        d = json.load(open(f))
        print(f, {k: (v["ms_per_token"], v["logits_equal_first_run"], v["error_word"], v["max_abs_diff_vs_first_run"]) for k, v in d["runs"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY

#!/bin/bash
# round 4, session 14: what cross-workgroup synchronisation costs (probe), the streamed decode schedule on agent-scope single-word accesses
# (no cache-wide write-back / invalidate): bit-equality on hardware, A/B on the 25.3 B model, timing-only ablations (no ticket / no waits / no sync)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 200 python tools/probes/stream_sync_costs.py > gpurun_out/r04_stream_sync_costs.json 2> gpurun_out/r04_stream_sync_costs.err
( time timeout 420 python -m pytest tests/test_gpu_model.py -m gpu -q -k "streamed" 2>&1 | tail -12 ) > gpurun_out/r04_s14_pytest.log 2>&1
timeout 420 python tools/probes/decode_stream_ab.py > gpurun_out/r04_decode_stream_ab2.json 2> gpurun_out/r04_decode_stream_ab2.err
for v in noticket nowait nosync; do
  timeout 300 python tools/probes/decode_stream_ab.py --lib=build/abl/libaria_decode_$v.so > gpurun_out/r04_decode_stream_ab2_$v.json 2> gpurun_out/r04_decode_stream_ab2_$v.err
done
cat gpurun_out/r04_stream_sync_costs.json; tail -3 gpurun_out/r04_stream_sync_costs.err; tail -5 gpurun_out/r04_s14_pytest.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_decode_stream_ab2*.json")):
    try:
        d = json.load(open(f))
        print(f, {k: (v["ms_per_token"], v["logits_equal_first_run"], v["error_word"]) for k, v in d["runs"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY

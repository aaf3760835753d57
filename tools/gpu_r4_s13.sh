#!/bin/bash
# round 4, session 13: the streamed decode schedule (one launch per token) on hardware: bit-equality with the 6-launch schedule at toy and
# Aria widths, the A/B on the 25.3 B model (plain + captured), the compiler-fence variant of the acquire, the generate record with it on
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 420 python -m pytest tests/test_gpu_model.py -m gpu -q -k "streamed or fused_schedule or decode_engine" 2>&1 | tail -12 ) > gpurun_out/r04_s13_pytest.log 2>&1
timeout 420 python tools/probes/decode_stream_ab.py --graph > gpurun_out/r04_decode_stream_ab.json 2> gpurun_out/r04_decode_stream_ab.err
timeout 300 python tools/probes/decode_stream_ab.py --lib=build/abl/libaria_decode_fence.so > gpurun_out/r04_decode_stream_ab_fence.json 2> gpurun_out/r04_decode_stream_ab_fence.err
ARIA_DECODE_STREAM=1 timeout 420 python bench.py --no-cpu-baseline --no-long64k --steps 3 --warmup 1 > gpurun_out/r04_s13_bench_stream.json 2> gpurun_out/r04_s13_bench_stream.err
tail -5 gpurun_out/r04_s13_pytest.log; cat gpurun_out/r04_decode_stream_ab.json; tail -2 gpurun_out/r04_decode_stream_ab.err; cat gpurun_out/r04_decode_stream_ab_fence.json; tail -2 gpurun_out/r04_decode_stream_ab_fence.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04_s13_bench_stream.json"))
    print(json.dumps(d.get("generate_config2"))[:900])
except Exception as e:
    print("bench:", e)
PY

#!/bin/bash
# round 4, session 5: decode with non-temporal weight loads (tree) against plain loads (build/abl/libaria_decode_plain.so), same box, A-B-A
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
cp aria_amd/libaria_hip.so /tmp/new.so
run() { timeout 300 python tools/generate_bench.py 2>/dev/null | tail -1 > gpurun_out/r04_decode_$1.json; python -c "import json; d=json.load(open('gpurun_out/r04_decode_$1.json')); print('$1', d['value'], d['decode_ms_per_token'], d['engine_call_ms'])"; }
run nt1
cp build/abl/libaria_decode_plain.so aria_amd/libaria_hip.so; run plain
cp /tmp/new.so aria_amd/libaria_hip.so; run nt2
( time timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "decode or sample or forward_variants" 2>&1 | tail -4 ) > gpurun_out/r04_s5_pytest.log 2>&1
tail -3 gpurun_out/r04_s5_pytest.log

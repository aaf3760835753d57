#!/bin/bash
# Round-3 session 7: the always-on hardware suite on the tree with the 6-launch decode schedule and the SwiGLU-backward GEMM epilogue
# (both have their own hardware cases in it), then the same-process A/B of the fused epilogue and the decode-step timing of both schedules.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 1000 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -30 ) > gpurun_out/s7_pytest.log 2>&1
timeout 240 python tools/dswiglu_ab.py > gpurun_out/s7_dswiglu_ab.json 2> gpurun_out/s7_dswiglu_ab.err
timeout 420 python tools/decode_bench.py > gpurun_out/s7_decode.json 2> gpurun_out/s7_decode.err
grep -E "passed|failed|error" gpurun_out/s7_pytest.log | tail -3; cat gpurun_out/s7_dswiglu_ab.json; tail -1 gpurun_out/s7_decode.json | cut -c1-600
tail -3 gpurun_out/s7_dswiglu_ab.err gpurun_out/s7_decode.err

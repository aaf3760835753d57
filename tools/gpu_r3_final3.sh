#!/bin/bash
# decode attention with 16 waves x 8 keys per lane group (one pass over a chat-length context): hardware cases, then config #2 generate
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 240 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -k "decode or sample or generate" 2>&1 | tail -6 ) > gpurun_out/f3_pytest.log 2>&1
timeout 240 python tools/generate_bench.py > gpurun_out/f3_generate.json 2> gpurun_out/f3_generate.err
grep -E "passed|failed" gpurun_out/f3_pytest.log; tail -1 gpurun_out/f3_generate.json | cut -c1-500; tail -2 gpurun_out/f3_generate.err

#!/bin/bash
# round 4, session 8: dK/dV kernel with the hoisted mask test / batched lse reads / one-step fragment prefetch against the previous build
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/probes/attn_bwd_r4_ab.py > gpurun_out/r04_attn_bwd_ab.json 2> gpurun_out/r04_attn_bwd_ab.err
( time timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullwidth.py -m gpu -q -x -k "attention" 2>&1 | tail -4 ) > gpurun_out/r04_s8_pytest.log 2>&1
cat gpurun_out/r04_attn_bwd_ab.json; tail -2 gpurun_out/r04_attn_bwd_ab.err; tail -3 gpurun_out/r04_s8_pytest.log

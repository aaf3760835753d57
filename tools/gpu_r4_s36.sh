#!/bin/bash
# round 4, session 36: the decode engine's hardware cases on the final decode.hip (bisection switches removed)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -k "streamed or decode or fused_schedule or sample" 2>&1 | tail -4 ) > gpurun_out/r04_s36_pytest.log 2>&1
tail -3 gpurun_out/r04_s36_pytest.log

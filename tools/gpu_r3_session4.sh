#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attention" 2>&1 | tail -6 ) > gpurun_out/s4_pytest_attn.log 2>&1
timeout 600 python tools/probes/attn_fwd_r3_ab.py > gpurun_out/s4_attn_fwd_ab.log 2>&1
tail -3 gpurun_out/s4_pytest_attn.log; grep "^{" gpurun_out/s4_attn_fwd_ab.log

#!/bin/bash
# round 4, session 33: the pipelined dK/dV kernel (ARIA_ATTN_DKDV=6) on hardware: bit-equality cases, A/B at 2 K / 16 K / 64 K
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "dkdv_pipelined" 2>&1 | tail -4 ) > gpurun_out/r04_s33_pytest.log 2>&1
timeout 200 python tools/probes/attn_dkdv6_ab.py > gpurun_out/r04_attn_dkdv6_ab.json 2> gpurun_out/r04_attn_dkdv6_ab.err
tail -2 gpurun_out/r04_s33_pytest.log; cat gpurun_out/r04_attn_dkdv6_ab.json; tail -2 gpurun_out/r04_attn_dkdv6_ab.err

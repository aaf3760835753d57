#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullwidth.py tests/test_gpu_ep.py -m gpu -q -k "recompute or ep or rccl" 2>&1 | tail -8 ) > gpurun_out/s6_pytest.log 2>&1
( time timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/s6_bench.json 2> gpurun_out/s6_bench.err
ARIA_RECOMPUTE_LEVEL=layer timeout 600 python bench.py --no-cpu-baseline --steps 2 2>/dev/null | grep "^{" > gpurun_out/s6_bench_level_layer.json
tail -3 gpurun_out/s6_pytest.log; grep "^{" gpurun_out/s6_bench.json | cut -c1-200

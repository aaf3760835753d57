#!/bin/bash
# round 4, session 12: dispatch kernels with a compile-time row width and the DPP wave sums in the norm kernels: parity cases, A/B, bench
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_ep.py -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/r04_s12_pytest.log 2>&1
timeout 300 python tools/probes/dispatch_ab.py > gpurun_out/r04_dispatch_ab.json 2> gpurun_out/r04_dispatch_ab.err
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04_s12_bench.json 2> gpurun_out/r04_s12_bench.err
tail -3 gpurun_out/r04_s12_pytest.log; cat gpurun_out/r04_dispatch_ab.json; tail -2 gpurun_out/r04_dispatch_ab.err; cut -c1-250 gpurun_out/r04_s12_bench.json

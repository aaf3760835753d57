#!/bin/bash
# round 4, session 11: the whole hardware suite on HEAD, bench.py (driver's line), bench.py --ep at world 1
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -16 ) > gpurun_out/r04_s11_pytest_full.log 2>&1
( time timeout 600 python bench.py > gpurun_out/r04_s11_bench.json 2> gpurun_out/r04_s11_bench.err ) 2> gpurun_out/r04_s11_bench.time
timeout 600 python bench.py --ep --steps 3 --warmup 1 --no-cpu-baseline --no-long64k > gpurun_out/r04_s11_bench_ep.json 2> gpurun_out/r04_s11_bench_ep.err
tail -6 gpurun_out/r04_s11_pytest_full.log; cut -c1-300 gpurun_out/r04_s11_bench.json; cut -c1-300 gpurun_out/r04_s11_bench_ep.json; tail -2 gpurun_out/r04_s11_bench_ep.err

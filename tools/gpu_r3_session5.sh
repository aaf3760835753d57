#!/bin/bash
# Round-3 GPU session 5: the complete always-on hardware suite (timed), the driver's bench line with its sub-records, bench --ep (world 1).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -40 ) > gpurun_out/s5_pytest.log 2>&1
( time timeout 900 python bench.py ) > gpurun_out/s5_bench.json 2> gpurun_out/s5_bench.err
( time timeout 600 python bench.py --ep --steps 3 --warmup 1 --no-cpu-baseline --no-long64k ) > gpurun_out/s5_bench_ep.json 2> gpurun_out/s5_bench_ep.err
tail -4 gpurun_out/s5_pytest.log; grep "^{" gpurun_out/s5_bench.json | cut -c1-300

#!/bin/bash
# Round-3 final GPU session: always-on hardware suite, the driver's bench line, its rocprofv3 kernel trace, smoke, then the two slow
# long-shape parity cases (T = 65 536 layer, 53 248-token prefill) on the final kernels.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -25 ) > gpurun_out/f_pytest.log 2>&1
( time timeout 900 python bench.py ) > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/f_smoke.log 2>&1
bash tools/gpu_prof_bench.sh r03f --no-long64k
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_long" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --long --steps 1 --warmup 1 --no-cpu-baseline ) > gpurun_out/prof_long.log 2>&1
db=$(find gpurun_out/prof_long -name '*.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" 25 > gpurun_out/kernel_stats_long_r03f.txt 2>&1; rm -rf gpurun_out/prof_long
( time ARIA_SLOW_TESTS=1 timeout 2400 python -m pytest tests/test_gpu_fullwidth.py -m gpu -q -k "T65536 or 53248" --durations=4 2>&1 | tail -12 ) > gpurun_out/f_pytest_slow.log 2>&1
grep -E "passed|failed" gpurun_out/f_pytest.log gpurun_out/f_pytest_slow.log; tail -1 gpurun_out/f_smoke.log; grep "^{" gpurun_out/f_bench.json | cut -c1-240

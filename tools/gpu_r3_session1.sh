#!/bin/bash
# Round-3 GPU session 1: L2-atomics probe, the driver's bench line (now with the long64k sub-record), bench --ep on a one-rank RCCL group,
# then the hardware parity suite incl. the new long-shape cases (S = 65 536 attention, T = 65 536 layer, 53 248-token prefill, EP on HIP).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 120 python tools/probes/l2_atomics.py > gpurun_out/s1_l2_atomics.log 2>&1
( time timeout 900 python bench.py ) > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err
( time timeout 600 python bench.py --ep --steps 3 --warmup 1 --no-cpu-baseline --no-long64k ) > gpurun_out/s1_bench_ep.json 2> gpurun_out/s1_bench_ep.err
( time timeout 2400 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -60 ) > gpurun_out/s1_pytest.log 2>&1
tail -5 gpurun_out/s1_pytest.log; tail -c 1500 gpurun_out/s1_bench.json

#!/bin/bash
# round 4, final session on HEAD: the whole hardware suite, the driver's bench line, smoke, the rocprofv3 kernel trace of the bench command,
# the PMC passes of the roofline kernel (fc1 + SwiGLU at KERNEL_REV r04b) and of the attention backward.  Ordered by importance.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -24 ) > gpurun_out/r04_final_pytest.log 2>&1
( time timeout 600 python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err ) 2> gpurun_out/r04_final_bench.time
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_final_smoke.log 2>&1
bash tools/gpu_prof_bench.sh r04f --no-long64k --no-inference-records
bash tools/gpu_r4_pmc.sh fc1 ${1:-attn_bwd} > gpurun_out/r04_final_pmc.log 2>&1
grep -E "passed|failed|error" gpurun_out/r04_final_pytest.log | tail -3; tail -1 gpurun_out/r04_final_smoke.log; cut -c1-400 gpurun_out/r04_final_bench.json
head -8 gpurun_out/kernel_stats_r04f.txt | cut -c1-80,112-175; ls gpurun_out/r04_pmc_fc1_*.csv

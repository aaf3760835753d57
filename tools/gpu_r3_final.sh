#!/bin/bash
# Round-3 final GPU session (13 GPU-minutes left): hardware cases of everything that changed after session 7's full suite (kernel cases,
# model cases incl. the decode engine, the Aria-width gptfast prefill), the driver's bench line, smoke, its rocprofv3 kernel trace,
# config #2 generate and config #4 prefill on the final kernels.  Ordered by importance: the budget may cut the tail.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 420 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullwidth.py -m gpu -q --durations=5 \
    -k "not fullwidth or prefill_gptfast_16384" 2>&1 | tail -15 ) > gpurun_out/f_pytest.log 2>&1
( time timeout 600 python bench.py ) > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/f_smoke.log 2>&1
bash tools/gpu_prof_bench.sh r03f --no-long64k
if [ "$1" = "all" ]; then bash tools/gpu_r3_final2.sh; fi
grep -E "passed|failed" gpurun_out/f_pytest.log; tail -1 gpurun_out/f_smoke.log; grep "^{" gpurun_out/f_bench.json | cut -c1-300

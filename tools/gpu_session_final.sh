#!/bin/bash
# Evidence session: hardware parity, the bench lines (default = config #3, --recompute, --long), kernel trace of the default command,
# PMC passes on the fc1 launch, config #2 / #4 tools.  Outputs under gpurun_out/ (copied into profiles/ by hand).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/f_pytest.log 2>&1
python bench.py --steps 10 --warmup 3 > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err
python bench.py --steps 5 --warmup 2 --recompute --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/f_bench_recompute.json
python bench.py --long --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/f_bench_long.json
bash tools/gpu_prof_bench.sh r02f
bash tools/gpu_pmc_fc1.sh
timeout 600 python tools/generate_bench.py > gpurun_out/f_generate.json 2> gpurun_out/f_generate.err
timeout 600 python tools/prefill_bench.py > gpurun_out/f_prefill.json 2> gpurun_out/f_prefill.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/f_smoke.log 2>&1
tail -3 gpurun_out/f_pytest.log; tail -1 gpurun_out/f_smoke.log

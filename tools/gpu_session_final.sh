cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > gpurun_out/f_pytest.log 2>&1
python bench.py --steps 10 --warmup 3 > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err
python bench.py --steps 5 --warmup 2 --recompute --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/f_bench_recompute.json
python bench.py --long --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/f_bench_long.json
bash tools/gpu_prof_bench.sh r02f
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_long" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --long --steps 1 --warmup 1 --no-cpu-baseline ) > gpurun_out/prof_long.log 2>&1
db=$(find gpurun_out/prof_long -name '*.db' | head -1); python tools/rocpd_stats.py "$db" 30 > gpurun_out/kernel_stats_long.txt 2>&1; rm -rf gpurun_out/prof_long
timeout 600 python tools/prefill_bench.py > gpurun_out/f_prefill.json 2> gpurun_out/f_prefill.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/f_smoke.log 2>&1
tail -3 gpurun_out/f_pytest.log; tail -1 gpurun_out/f_smoke.log

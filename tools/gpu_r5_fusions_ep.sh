cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
export ARIA_PYTEST_FILES="tests/test_gpu_kernels.py tests/test_gpu_ep.py tests/test_gpu_model.py"
bash tools/gpu_session.sh r05s5 "pytest=router_fused or qkv_rope_hf or expert_parallel or ep or lora"
F="--no-long64k --no-inference-records --no-cpu-baseline --no-lora-record --steps 6 --warmup 2"
python bench.py $F > gpurun_out/r05s5_bench_fused.json 2> gpurun_out/r05s5_bench_fused.err
ARIA_FUSE_ROUTER=0 ARIA_FUSE_QKV_ROPE=0 python bench.py $F > gpurun_out/r05s5_bench_unfused.json 2> gpurun_out/r05s5_bench_unfused.err
python bench.py $F > gpurun_out/r05s5_bench_fused2.json 2> gpurun_out/r05s5_bench_fused2.err
ARIA_EP_CHUNKS=1 python bench.py --ep $F > gpurun_out/r05s5_bench_ep_c1.json 2> gpurun_out/r05s5_bench_ep_c1.err
ARIA_EP_CHUNKS=2 python bench.py --ep $F > gpurun_out/r05s5_bench_ep_c2.json 2> gpurun_out/r05s5_bench_ep_c2.err
ARIA_EP_CHUNKS=4 python bench.py --ep $F > gpurun_out/r05s5_bench_ep_c4.json 2> gpurun_out/r05s5_bench_ep_c4.err
tail -6 gpurun_out/r05s5_pytest.log
python - <<'PY'
import json
for f in ("fused","unfused","fused2","ep_c1","ep_c2","ep_c4"):
    try:
        d=json.load(open(f"gpurun_out/r05s5_bench_{f}.json")); print(f, d["ms_per_step"], d["roofline"]["achieved"], d["config"]["parallelism"])
    except Exception as e:
        print(f, "ERR", e, open(f"gpurun_out/r05s5_bench_{f}.err").read()[-600:])
PY

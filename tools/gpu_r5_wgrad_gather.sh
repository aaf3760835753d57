cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
export ARIA_PYTEST_FILES="tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullwidth.py"
bash tools/gpu_session.sh r05s8 "pytest=gathered_rows or without_permuted or decoder_layer_aria_width_T4096 or two_decoder_layers or qkv_rope_hf or lora_fused_node or T65536_recompute"
F="--no-long64k --no-inference-records --no-cpu-baseline --no-lora-record --steps 6 --warmup 2"
for i in 1 2; do
  python bench.py $F > gpurun_out/r05s8_bench_gather$i.json 2> gpurun_out/r05s8_bench_gather$i.err
  ARIA_FUSE_WGRAD_GATHER=0 python bench.py $F > gpurun_out/r05s8_bench_permute$i.json 2> gpurun_out/r05s8_bench_permute$i.err
done
tail -6 gpurun_out/r05s8_pytest.log
python - <<'PY'
import json
for f in ("gather1","permute1","gather2","permute2"):
    try:
        d=json.load(open(f"gpurun_out/r05s8_bench_{f}.json")); print(f, d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["kernel"][:60], d["roofline"]["launches_timed"])
    except Exception as e:
        print(f, "ERR", e, open(f"gpurun_out/r05s8_bench_{f}.err").read()[-800:])
PY

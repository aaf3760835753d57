cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
export ARIA_PYTEST_FILES="tests/test_gpu_kernels.py tests/test_gpu_model.py"
bash tools/gpu_session.sh r05s10 "pytest=dswiglu or lora or gathered_rows or without_permuted or gemm_v3 or grouped_gemm"
python tools/dswiglu_ab.py > gpurun_out/r05s10_dswiglu_ab.json 2> gpurun_out/r05s10_dswiglu_ab.err
F="--no-long64k --no-inference-records --no-cpu-baseline --steps 6 --warmup 2"
python bench.py $F > gpurun_out/r05s10_bench.json 2> gpurun_out/r05s10_bench.err
tail -4 gpurun_out/r05s10_pytest.log; cat gpurun_out/r05s10_dswiglu_ab.json | cut -c1-600
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05s10_bench.json")); print(d["ms_per_step"], d["roofline"]["achieved"], {k:v for k,v in d.get("lora_config",{}).items() if k in ("ms_per_step","frozen_base_fwd_dgrad_ms","over_frozen_base","recipe_grad_checkpointing_ms")}, d["lora_config"]["roofline"]["achieved"])
PY

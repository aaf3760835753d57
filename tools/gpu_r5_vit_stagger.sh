cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=r05s13
export ARIA_PYTEST_FILES="tests/test_gpu_kernels.py"
bash tools/gpu_session.sh $tag "pytest=staggered or hd72 or dswiglu or masked_tiles"
python tools/probes/attn_hd72_stagger_ab.py > gpurun_out/${tag}_attn_hd72_stagger_ab.json 2> gpurun_out/${tag}_attn_hd72_stagger_ab.err
cut -c1-900 gpurun_out/${tag}_attn_hd72_stagger_ab.json; tail -2 gpurun_out/${tag}_attn_hd72_stagger_ab.err
python tools/dswiglu_ab.py > gpurun_out/${tag}_dswiglu_ab.json 2> gpurun_out/${tag}_dswiglu_ab.err; cut -c1-400 gpurun_out/${tag}_dswiglu_ab.json
best=$(python -c "import json;print(json.load(open('gpurun_out/${tag}_attn_hd72_stagger_ab.json'))['best_mode_vit'])" 2>/dev/null || echo 1)
[ "$best" = "0" ] && best=1
F="--no-long64k --no-inference-records --no-cpu-baseline --no-lora-record --steps 6 --warmup 2"
for r in 1 2; do
  for m in 0 $best; do
    ARIA_ATTN_HD72_STAGGER=$m python bench.py $F > gpurun_out/${tag}_bench_m${m}_$r.json 2> gpurun_out/${tag}_bench_m${m}_$r.err
    python -c "import json;d=json.load(open('gpurun_out/${tag}_bench_m${m}_$r.json'));print('mode',$m,'round',$r,d['ms_per_step'],d['roofline']['achieved'])"
  done
done

#!/bin/bash
# round 4, session 6: K2 (row gather in fc1's loader) and K7 (wqkv + RoPE + cache write) on hardware: kernel cases, the prefill parity
# cases, config #4 prefill with the fusions on / off (same box), config #2 generate
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullwidth.py -m gpu -q -x -k "row_gather or rope_and_cache or prefill or forward_variants" 2>&1 | tail -6 ) > gpurun_out/r04_s6_pytest.log 2>&1
timeout 300 python tools/prefill_bench.py > gpurun_out/r04_prefill_fused.json 2> gpurun_out/r04_prefill.err
ARIA_FUSE_GATHER=0 timeout 300 python tools/prefill_bench.py > gpurun_out/r04_prefill_no_k2.json 2>> gpurun_out/r04_prefill.err
ARIA_FUSE_QKV_ROPE=0 timeout 300 python tools/prefill_bench.py > gpurun_out/r04_prefill_no_k7.json 2>> gpurun_out/r04_prefill.err
ARIA_FUSE_GATHER=0 ARIA_FUSE_QKV_ROPE=0 ARIA_ATTN_FWD=2 timeout 300 python tools/prefill_bench.py > gpurun_out/r04_prefill_r03_path.json 2>> gpurun_out/r04_prefill.err
timeout 300 python tools/prefill_bench.py > gpurun_out/r04_prefill_fused2.json 2>> gpurun_out/r04_prefill.err
tail -4 gpurun_out/r04_s6_pytest.log; for f in fused no_k2 no_k7 r03_path fused2; do echo $f $(cut -c1-200 gpurun_out/r04_prefill_$f.json); done; tail -2 gpurun_out/r04_prefill.err

#!/bin/bash
# round 4, session 1: the always-on target-shape parity cases on the final kernels, bench.py with the new sub-records, the MFMA shape /
# co-issue probe, PMC passes of the three roofline kernels on HEAD
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_fullwidth.py -m gpu -q -x --durations=12 -k "oracle_on_device or T65536 or 53248 or beyond_2g" 2>&1 | tail -30 ) > gpurun_out/r04_s1_pytest_long.log 2>&1
( time timeout 600 python bench.py > gpurun_out/r04_s1_bench.json 2> gpurun_out/r04_s1_bench.err ) 2> gpurun_out/r04_s1_bench.time
timeout 120 ./build/abl/mfma_shapes > gpurun_out/r04_mfma_shapes.jsonl 2>&1
timeout 900 bash tools/gpu_r4_pmc.sh fc1 vit_fwd attn_bwd > gpurun_out/r04_s1_pmc.log 2>&1
tail -5 gpurun_out/r04_s1_pytest_long.log; cut -c1-600 gpurun_out/r04_s1_bench.json; tail -3 gpurun_out/r04_s1_bench.err; cat gpurun_out/r04_mfma_shapes.jsonl | cut -c1-200

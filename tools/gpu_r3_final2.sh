#!/bin/bash
# Round-3 last GPU minutes: config #2 generate and config #4 prefill on the final kernels (6-launch decode schedule, fused gate / up +
# SwiGLU prefill on the model.pth wire format).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 240 python tools/generate_bench.py > gpurun_out/f_generate.json 2> gpurun_out/f_generate.err
timeout 240 python tools/prefill_bench.py --runs 1 > gpurun_out/f_prefill.json 2> gpurun_out/f_prefill.err
tail -1 gpurun_out/f_generate.json | cut -c1-500; tail -1 gpurun_out/f_prefill.json | cut -c1-300; tail -2 gpurun_out/f_generate.err gpurun_out/f_prefill.err

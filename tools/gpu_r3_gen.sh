#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 240 python tools/generate_bench.py > gpurun_out/f4_generate.json 2> gpurun_out/f4_generate.err
tail -1 gpurun_out/f4_generate.json | cut -c1-600; tail -2 gpurun_out/f4_generate.err

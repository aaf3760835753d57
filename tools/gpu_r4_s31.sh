#!/bin/bash
# round 4, session 31 (after the final session, spare minutes): the no-waiting streamed decode variant (tools/probes/src/decode_stream_sentinel.patch,
# built as a variant library) with readers polling the data itself and fewer resident workgroups per CU
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 100 python build/sentinel/tools/probes/decode_stream_ab.py --sweep --lib=build/abl/libaria_decode_sentinel.so > gpurun_out/r04_decode_stream_sweep.json 2> gpurun_out/r04_decode_stream_sweep.err
grep -v amdgpu.ids gpurun_out/r04_decode_stream_sweep.err | cut -c1-400 | tail -12

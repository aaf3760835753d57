#!/bin/bash
# round 4, session 15: GEMV-shaped streaming rate vs bytes in flight per CU (probe)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python tools/probes/gemv_stream_rate.py > gpurun_out/r04_gemv_stream_rate.json 2> gpurun_out/r04_gemv_stream_rate.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_gemv_stream_rate.json"))
for r in d["runs"]:
    print(r["rows_in_registers"], r["rows_through_lds"], r["workgroups_per_cu_by_lds"], r["kb_in_flight_per_cu"], r["TB_s"])
PY
tail -3 gpurun_out/r04_gemv_stream_rate.err

#!/bin/bash
# L2 behaviour of the v3 GEMM (separate PMC passes, --kernel-trace only): hits / misses, fabric-side fetches, read latency counters
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TCC|TCP)_[A-Z0-9_]*(LATENCY|LEVEL|STALL|BUSY|REQ)[A-Z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/pmc_avail.txt
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" FETCH_SIZE "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcl2_$tag -o p -- python "$GRAFT_REPO_ROOT/tools/probes/pmc_dense_target.py" ) > gpurun_out/pmcl2_$tag.log 2>&1
  f=$(find /tmp/pmcl2_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" > gpurun_out/pmcl2_$tag.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    if 'gemm3' not in r['Kernel_Name']: continue
    key = (r['Kernel_Name'][:40], r['Grid_Size'], r['Counter_Name'])
    agg.setdefault(key, []).append(float(r['Counter_Value']))
for k, v in agg.items(): print(k, len(v), sum(v) / len(v))
PY
done
tail -3 gpurun_out/pmcl2_*.log | head -40

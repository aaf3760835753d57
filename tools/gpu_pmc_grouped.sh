cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( cd /tmp && timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmcg -o p -- python "$GRAFT_REPO_ROOT/tools/probes/pmc_grouped_target.py" ) > gpurun_out/pmcg.log 2>&1
f=$(find /tmp/pmcg -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm3' in r['Kernel_Name']]
d = {}
for r in rows: d.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
for k in sorted(d): print(k, d[k], round(d[k]['TCC_HIT_sum'] / (d[k]['TCC_HIT_sum'] + d[k]['TCC_MISS_sum']), 3))
PY

#!/bin/bash
# r06 probe of the 4-wave / 128 x 128-per-wave K loop (tools/probes/src/gemm6_4wave.hip) beside the product's gemm3 kernels, ONE process per
# shape: plain timing first, then the same binary under rocprofv3 --pmc (counters in their own passes, --kernel-trace only) for
# SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / MFMA busy / GRBM_GUI_ACTIVE (effective clock = GRBM_GUI_ACTIVE / kernel duration).
#   gpurun -- 'bash tools/probes/gemm6_probe.sh <tag>'   ->  gpurun_out/<tag>_gemm6_*.json / *_pmc_*.csv ; summary: tools/probes/gemm6_summary.py
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out build/abl
tag=${1:-r06}
BIN=build/abl/gemm6_4wave
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iaria_amd/csrc -Iinclude tools/probes/src/gemm6_4wave.hip -o $BIN -ldl
for shape in "16384 3328 2560 0" "16384 3328 2560 1" "8192 8192 8192 0" "16384 2560 3328 0" "78336 4352 1152 0" "78336 1280 4352 0" "16384 7680 2560 0"; do
  name=$(echo $shape | tr ' ' 'x')
  timeout 300 $BIN $shape ${ITERS:-100} 3 > gpurun_out/${tag}_gemm6_$name.json 2> gpurun_out/${tag}_gemm6_$name.err
  cut -c1-900 gpurun_out/${tag}_gemm6_$name.json
done
for shape in "16384 3328 2560 0" "16384 3328 2560 1"; do
  name=$(echo $shape | tr ' ' 'x')
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16" "GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum"; do
    ctag=$(echo $c | cut -d' ' -f1)
    rm -rf /tmp/g6_$ctag
    ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/g6_$ctag -o p -- "$GRAFT_REPO_ROOT/$BIN" $shape 3 1 ) > gpurun_out/${tag}_gemm6_pmc_${name}_$ctag.log 2>&1
    f=$(find /tmp/g6_$ctag -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && grep -E 'Counter_Name|gemm' "$f" | cut -c1-900 > gpurun_out/${tag}_gemm6_pmc_${name}_$ctag.csv
    k=$(find /tmp/g6_$ctag -name '*kernel_trace.csv' | head -1)
    [ -n "$k" ] && grep -E 'Kernel_Name|gemm' "$k" | cut -c1-700 > gpurun_out/${tag}_gemm6_trace_${name}_$ctag.csv
  done
done
ls gpurun_out/${tag}_gemm6_* | wc -l

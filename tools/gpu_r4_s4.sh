#!/bin/bash
# round 4, session 4: attn_fwd3 variants (hd 72: 12 waves + LDS-DMA against the pipelined 8-wave form and v2), dK/dV ablation
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/probes/attn_fwd3_ab.py > gpurun_out/r04_attn_fwd3_ab2.json 2> gpurun_out/r04_attn_fwd3_ab2.err
timeout 900 python tools/probes/attn_dkdv_ablate.py > gpurun_out/r04_attn_dkdv_ablate.json 2> gpurun_out/r04_attn_dkdv_ablate.err
( time timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "attention or vit or decoder or lm_golden" 2>&1 | tail -5 ) > gpurun_out/r04_s4_pytest.log 2>&1
cat gpurun_out/r04_attn_fwd3_ab2.json; tail -2 gpurun_out/r04_attn_fwd3_ab2.err; cat gpurun_out/r04_attn_dkdv_ablate.json; tail -2 gpurun_out/r04_attn_dkdv_ablate.err; tail -4 gpurun_out/r04_s4_pytest.log

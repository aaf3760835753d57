cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_lora" -o p -- python "$GRAFT_REPO_ROOT/tools/lora_profile.py" 4 ) > gpurun_out/r05s3_lora_prof.log 2>&1
db=$(find gpurun_out/prof_lora -name '*.db' | head -1)
python tools/rocpd_stats.py "$db" 45 > gpurun_out/r05s3_lora_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_lora
grep '^{' gpurun_out/r05s3_lora_prof.log; cut -c1-150,112-190 gpurun_out/r05s3_lora_kernel_stats.txt | head -50

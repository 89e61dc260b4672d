cd /tmp; export TMPDIR=/tmp
rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_hf1080 -o hf --output-format csv -- $GRAFT_REPO_ROOT/lumahdrv_amd/bin/facade_hostfed 1920 1080 48 > $GRAFT_REPO_ROOT/gpurun_out/prof_hf1080.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_hf1080.log
ls $GRAFT_REPO_ROOT/gpurun_out/prof_hf1080/
head -30 $GRAFT_REPO_ROOT/gpurun_out/prof_hf1080/hf_hip_api_stats.csv
head -12 $GRAFT_REPO_ROOT/gpurun_out/prof_hf1080/hf_kernel_stats.csv
head -8 $GRAFT_REPO_ROOT/gpurun_out/prof_hf1080/hf_memory_copy_stats.csv

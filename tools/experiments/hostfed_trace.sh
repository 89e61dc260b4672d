# tools/experiments/hostfed_trace.sh [w h] -- HIP API / copy / kernel timeline of the drop-in call on host frames (facade_hostfed)
W=${1:-3840}; H=${2:-2160}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_hf${H}
rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --stats -d $OUT -o hf --output-format csv -- $GRAFT_REPO_ROOT/lumahdrv_amd/bin/facade_hostfed $W $H 24 > $OUT.log 2>&1
tail -1 $OUT.log
head -30 $OUT/hf_hip_api_stats.csv
head -12 $OUT/hf_kernel_stats.csv
head -8 $OUT/hf_memory_copy_stats.csv

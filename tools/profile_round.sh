#!/bin/bash
# tools/profile_round.sh <tag> [workload [width height frames_per_launch]] -- run on the GPU box (via gpurun).  Collects, under
# gpurun_out/prof_<tag>_<wl>/ (default shape 3840x2160 x 20; BASELINE configs[3] is `log12_luv 7680 4320 5`):
#   stats/   rocprofv3 --kernel-trace --stats of `python bench.py --steps 10 --warmup 2` (the default bench command,
#            shorter), whose per-kernel average must agree with bench.py's own hipEvent timing;
#   pmc_*/   PMC counters of tools/prof_driver.py, one rocprofv3 run per counter group (--pmc is never combined
#            with other tracing domains; FETCH_SIZE and WRITE_SIZE do not fit one pass; 8 SQ counters per pass).
set -u
TAG=${1:-r02}
WL=${2:-pq11_luv}
W=${3:-3840}; H=${4:-2160}; B=${5:-20}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_$WL
mkdir -p $OUT
P="rocprofv3 --kernel-trace --output-format csv"
# --no-float-inputs: the float-input legs of the YCbCr workload run their first launches (and every probe) on the SAME half-input
# kernel on data it is slow for; without the flag that kernel's average in the stats below is no longer the headline launch's
BARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-other-workloads --no-facade-hostfed --no-placement-off --no-float-inputs --min-seconds 0.3 --workload $WL --width $W --height $H --frames-per-step $B"
echo "{\"width\": $W, \"height\": $H, \"frames\": $B}" > $OUT/shape.json
$P --stats -d $OUT/stats -o bench -- python bench.py $BARGS > $OUT/bench_under_rocprof.log 2>&1
python bench.py $BARGS > $OUT/bench_plain.log 2>&1
# the same K launches back to back on ONE stream: per-kernel durations in this trace are directly comparable with bench.py's
# kernel_ms_ordered (with two lanes two launches are in flight and each one's own duration is about twice the window / K)
$P --stats -d $OUT/stats_ordered -o bench -- python bench.py $BARGS --lanes 0 > $OUT/bench_ordered_under_rocprof.log 2>&1
$P --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_fetch -o p -- python tools/prof_driver.py 3 $WL $W $H $B > $OUT/pmc_fetch.log 2>&1
$P --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o p -- python tools/prof_driver.py 3 $WL $W $H $B > $OUT/pmc_write.log 2>&1
$P --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmc_inst -o p -- python tools/prof_driver.py 3 $WL $W $H $B > $OUT/pmc_inst.log 2>&1
$P --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/pmc_wait -o p -- python tools/prof_driver.py 3 $WL $W $H $B > $OUT/pmc_wait.log 2>&1
$P --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT -d $OUT/pmc_mix1 -o p -- python tools/prof_driver.py 3 $WL $W $H $B > $OUT/pmc_mix1.log 2>&1
$P --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F64 -d $OUT/pmc_mix2 -o p -- python tools/prof_driver.py 3 $WL $W $H $B > $OUT/pmc_mix2.log 2>&1
tail -n 1 $OUT/bench_plain.log | cut -c1-300
ls $OUT

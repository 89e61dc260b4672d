#!/bin/bash
# encode, 3 workgroups per CU (B: LUMAHIP_GRID_ENC=768) against the default 8 (A), same build, one process (tools/bench/ab_inproc.py)
export LUMAHIP_TUNING=1   # the LUMAHIP_* overrides are honoured only under this gate
L=lumahdrv_amd/lib/liblumahip.so
for cfg in "pq11_luv 2" "pq11_luv 3" "pq11_rgb 2" "pq11_rgb 3" "pq8_luv 0" "pq8_luv 1"; do
  set -- $cfg
  AB_ENV_B=LUMAHIP_GRID_ENC=${GRID:-768} AB_PROFILE=$2 AB_BATCHES=8 AB_ITERS=30 timeout 200 python tools/bench/ab_inproc.py $L $L $1 40 2>/dev/null | tail -1 | sed "s/^/[$1 profile $2] /"
done

#!/bin/bash
A=lumahdrv_amd/lib_old/liblumahip.so; B=lumahdrv_amd/lib/liblumahip.so
for cfg in "pq11_luv 2" "pq11_luv 3" "pq11_rgb 3" "pq11_rgb 2" "pq8_luv 0" "pq8_luv 1" "pq10_ycbcr 2"; do
  set -- $cfg
  AB_DIRECTION=1 AB_PROFILE=$2 AB_BATCHES=8 timeout 200 python tools/bench/ab_inproc.py $A $B $1 40 2>/dev/null | tail -4 | sed "s/^/[$1 profile $2] /" | grep -E "A = |B - A"
done

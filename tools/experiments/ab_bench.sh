#!/bin/bash
# same-box sustained A/B of two library builds through bench.py (headline workload only)
for r in 1 2 3; do
  for l in ${ORDER:-lib_old lib}; do
    LUMAHIP_LIB=$PWD/lumahdrv_amd/$l/liblumahip.so python bench.py --no-cpu-baseline --no-other-workloads --min-seconds 3 --max-repeats 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$l', 'ms/step', d['ms_per_step'], 'min', d['ms_per_step_min'], 'max', d['ms_per_step_max'], 'frac', r['frac'], 'probe', r['traffic_only_ms'], 'decode', d['decode_mpix_s'])"
  done
done

#!/bin/bash
# bench.py (headline workload only, placement auto) under different launch configurations, round-robin twice
export LUMAHIP_TUNING=1   # the LUMAHIP_* overrides are honoured only under this gate
run() {
  env "$@" python bench.py --no-cpu-baseline --no-other-workloads --min-seconds 1.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-44s ms/step %.4f frac %.4f probe %.4f decode %.0f y_apart %s' % ('$*', d['ms_per_step'], r['frac'], r['traffic_only_ms'], d['decode_mpix_s'], d['placement'].get('probe_ms',{}).get('chosen_layout_y_apart')))"
}
for r in 1 2; do
  run X=1
  run LUMAHIP_BLOCKS_PER_CU=6
  run LUMAHIP_BLOCKS_PER_CU=12
  run LUMAHIP_BLOCK=512
  run LUMAHIP_BLOCK=512 LUMAHIP_BLOCKS_PER_CU=8
  run LUMAHIP_BLOCK=128 LUMAHIP_BLOCKS_PER_CU=16
done

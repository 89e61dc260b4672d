#!/bin/bash
# persistent-workgroup count sweep of the encode and decode kernels through bench.py (headline workload, placement auto)
export LUMAHIP_TUNING=1   # the LUMAHIP_* overrides are honoured only under this gate
run() {
  env "$@" python bench.py --no-cpu-baseline --no-other-workloads --min-seconds 0.8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-44s encode %.0f (probe %.4f)  decode %.0f' % ('$*', d['value'], r['traffic_only_ms'], d['decode_mpix_s']))"
}
for g in 512 640 768 896 1024 1152 1280 1408 1536 1792 2048 2304 2560 3072 4096; do
  run LUMAHIP_GRID_ENC=$g LUMAHIP_GRID_DEC=$g
done

#!/bin/bash
export LUMAHIP_TUNING=1   # the LUMAHIP_* overrides are honoured only under this gate
for r in 1 2; do
for e in "X=1" "LUMAHIP_PROBE_DEPTH=2" "LUMAHIP_PROBE_DEPTH=4" "LUMAHIP_PROBE_DEPTH=2 LUMAHIP_GRID_ENC=512" "LUMAHIP_PROBE_DEPTH=4 LUMAHIP_GRID_ENC=512" "LUMAHIP_PROBE_DEPTH=3 LUMAHIP_GRID_ENC=768"; do
  env $e python bench.py --no-cpu-baseline --no-other-workloads --min-seconds 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-48s encode %.0f kernel_ms %.4f probe %.4f' % ('$e', d['value'], r['kernel_ms'], r['traffic_only_ms']))"
done; done

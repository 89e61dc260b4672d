#!/bin/bash
export LUMAHIP_TUNING=1   # the LUMAHIP_* overrides are honoured only under this gate
run() {
  env "$@" python bench.py --no-cpu-baseline --no-other-workloads --min-seconds 1.0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-44s encode %.0f  decode %.0f  roundtrip %.0f' % ('$*', d['value'], d['decode_mpix_s'], d['roundtrip_mpix_s']))"
}
for r in 1 2; do
  for n in 3 4 5 6 7 8; do run LUMAHIP_BLOCKS_PER_CU=$n; done
  run LUMAHIP_BLOCK=512 LUMAHIP_BLOCKS_PER_CU=2
  run LUMAHIP_BLOCK=512 LUMAHIP_BLOCKS_PER_CU=3
  run LUMAHIP_BLOCK=1024 LUMAHIP_BLOCKS_PER_CU=1
done

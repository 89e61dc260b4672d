#!/bin/bash
for r in 1 2; do
for B in 10 20 40; do
  python bench.py --placement off --frames-per-step $B --steps $((500/B)) --no-cpu-baseline --no-other-workloads --min-seconds 1.0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('B=$B  value %.0f  ms/step %.4f  kernel_ms %.4f  isolated %.4f  probe %.4f  kernel/probe %.3f  decode %.0f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_isolated_launch'], r['traffic_only_ms'], r['kernel_ms']/r['traffic_only_ms'], d['decode_mpix_s']))"
done; done

#!/bin/bash
for m in auto off auto off; do
  
  timeout 600 python bench.py --placement $m --no-cpu-baseline > gpurun_out/bench_pl_$m.json 2> gpurun_out/bench_pl_$m.err
  
  python - "$m" <<'EOF'
import json, sys
m = sys.argv[1]
try:
    d = json.load(open("gpurun_out/bench_pl_%s.json" % m))
except Exception as e:
    print(m, "no json", e); print(open("gpurun_out/bench_pl_%s.err" % m).read()[-1500:]); sys.exit(0)
r = d["roofline"]
print(m, d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "probe", r["traffic_only_ms"], "decode", d["decode_mpix_s"], "rt", d["roundtrip_mpix_s"], {k: (v["value"], v["roofline"].get("frac")) for k, v in d["other_workloads"].items()})
print("   ", d["placement"])
EOF
done

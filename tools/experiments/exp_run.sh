#!/bin/bash
# Interleaved A/B runner for the tools/experiments/exp_encode.hip variants: ROUNDS passes over "exe:args" pairs so that clock /
# thermal drift on the box is common-mode; prints the median over rounds of each case's per-run median.
#   tools/experiments/exp_run.sh 5 "base:256 8" "base:512 4" ...
ROUNDS=$1; shift
TMP=$(mktemp -d)
for r in $(seq $ROUNDS); do
  i=0
  for spec in "$@"; do
    exe=${spec%%:*}; args=${spec#*:}
    lumahdrv_amd/bin/exp_$exe $args | sed "s/^/[$exe $args] /" >> $TMP/$i.log
    i=$((i+1))
  done
done
python3 - "$TMP" <<'PY'
import sys, glob, re, statistics, collections
for f in sorted(glob.glob(sys.argv[1] + "/*.log"), key=lambda p: int(p.split("/")[-1][:-4])):
    d = collections.OrderedDict()
    for line in open(f):
        m = re.match(r"\[(.*?)\] (\S+-fed).*med ([0-9.]+) min ([0-9.]+)", line)
        if m:
            d.setdefault((m.group(1), m.group(2)), []).append((float(m.group(3)), float(m.group(4))))
    for (tag, case), v in d.items():
        print("%-22s %-9s median-of-medians %.4f  best-min %.4f  (n=%d)" % (tag, case, statistics.median(x[0] for x in v), min(x[1] for x in v), len(v)))
PY
rm -rf $TMP

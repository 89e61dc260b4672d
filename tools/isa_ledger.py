#!/usr/bin/env python3
"""tools/isa_ledger.py <file.s> <mangled-kernel-substring> [--blocks] -- instruction ledger of one kernel of a hipcc --save-temps
listing: instructions per class (fp64 / fp32 / conversions / integer / compare+select / LDS / global memory / scalar / other),
for the whole kernel and per basic block, so that the hot loop can be read off and priced with the issue costs of
profiles/r02_valu_issue_rates.txt (wave64 cycles: fp32 / int32 2, fp64 4, cvt / cmp / cndmask / min / max / med3 4, rcp 8)."""
import collections
import re
import sys


def classify(op):
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_branch", "s_cbranch", "s_setprio", "s_sleep")):
        return "ctl"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans"
    if op.startswith(("v_cmp", "v_cndmask", "v_min", "v_max", "v_med3", "v_minimum", "v_maximum")):
        return "cmp/sel/minmax"
    if re.search(r"_f64(_|$)", op):
        return "fp64"
    if re.search(r"_f32(_|$)", op) or op.startswith(("v_fma_", "v_mul_f", "v_add_f", "v_sub_f", "v_fmac", "v_fmaak", "v_fmamk", "v_div_", "v_ldexp", "v_frexp", "v_fract", "v_floor", "v_trunc", "v_rndne", "v_ceil")):
        return "fp32"
    if op.startswith(("v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr", "v_swap", "v_perm", "v_bfi", "v_alignbit", "v_pk_mov")):
        return "move/perm"
    if op.startswith("v_"):
        return "int"
    return "other"


COST = {"fp64": 4, "fp32": 2, "int": 2, "cvt": 4, "cmp/sel/minmax": 4, "trans": 8, "move/perm": 2}


def main():
    path, key = sys.argv[1], sys.argv[2]
    blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(("E:", ")")) or (l.startswith("_Z") and key in l and ":" in l))
    tot = collections.Counter()
    per = collections.OrderedDict()
    cur = "entry"
    ops = collections.Counter()
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end") or t.startswith("s_endpgm"):
            if t.startswith("s_endpgm"):
                tot["ctl"] += 1
            break
        m = re.match(r"^(\.LBB[0-9_]+):", t)
        if m:
            cur = m.group(1)
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        c = classify(op)
        tot[c] += 1
        per.setdefault(cur, collections.Counter())[c] += 1
        ops[op] += 1
    name = lines[start].split(":")[0]
    print("kernel", name)
    valu = sum(v for k, v in tot.items() if k in COST)
    cyc = sum(COST[k] * v for k, v in tot.items() if k in COST)
    print("whole kernel: %d instructions; VALU %d (%d issue cycles at the measured costs)" % (sum(tot.values()), valu, cyc))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print("  %-16s %5d" % (k, v))
    if blocks:
        print("basic blocks with >= 40 instructions:")
        for b, cnt in per.items():
            n = sum(cnt.values())
            if n >= 40:
                v = sum(x for k, x in cnt.items() if k in COST)
                cy = sum(COST[k] * x for k, x in cnt.items() if k in COST)
                print("  %-12s %5d instr, VALU %5d, %6d issue cycles: %s" % (b, n, v, cy, ", ".join("%s %d" % kv for kv in sorted(cnt.items(), key=lambda kv: -kv[1]))))
    if "--ops" in sys.argv:
        for op, n in ops.most_common(60):
            print("  %-28s %5d  [%s]" % (op, n, classify(op)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/summarize_profile.py <tag> [workload] -- turn gpurun_out/prof_<tag>_<workload>/ (tools/profile_round.sh) into
the committed artefacts: profiles/<tag>_<wl>_kernel_stats.csv, profiles/<tag>_<wl>_pmc_summary.md and one entry per
workload in profiles/traffic_latest.json (per-launch HBM bytes -> bench.py's roofline.traffic) and
profiles/valu_mix_latest.json (VALU instruction mix -> bench.py's VALU roofline).  Every entry carries the SHA of the
kernel sources it was captured from (lumahdrv_amd.capi.kernel_source_sha) and the commit; bench.py ignores entries whose
SHA differs from the sources it is running.

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE tallies the 128-B
requests of wide (16 B/lane) coalesced loads at 64 B (MI355X_MICROARCH.md, HBM section), hence the factor 2 on
the read side of the ENCODE kernel, whose pixel reads are all 16 B/lane.  Calibration inside the same run:
lh::k_synth writes a known byte count with 4-B stores (WRITE_SIZE ratio printed), and the decode kernel's
reads are 8-B / 4-B per lane (factor reported as measured/algorithmic, not assumed).

VALU issue cycles = sum over instruction classes of PMC count x issue cost per wave64 instruction measured with
tools/valu_bench.hip on MI355X (profiles/r02_valu_issue_rates.txt): fp32 add/mul/fma and int32 2 cycles, fp64 4,
conversions 4, fp32 transcendentals 8, everything the class counters do not name (compares, selects, min/max/med3,
moves, permutes, 64-bit integer) 4 -- an upper estimate for that remainder (moves issue in 2).
"""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

COST = {"SQ_INSTS_VALU_ADD_F32": 2, "SQ_INSTS_VALU_MUL_F32": 2, "SQ_INSTS_VALU_FMA_F32": 2, "SQ_INSTS_VALU_INT32": 2,
        "SQ_INSTS_VALU_ADD_F64": 4, "SQ_INSTS_VALU_MUL_F64": 4, "SQ_INSTS_VALU_FMA_F64": 4, "SQ_INSTS_VALU_CVT": 4,
        "SQ_INSTS_VALU_INT64": 4, "SQ_INSTS_VALU_TRANS_F32": 8, "SQ_INSTS_VALU_TRANS_F64": 16}
OTHER_COST = 4


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def update_json(path, wl, entry):
    try:
        d = json.load(open(path))
        if "workload" in d:      # round-1 single-entry format
            d = {}
    except Exception:
        d = {}
    d[wl] = entry
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)


def main():
    from lumahdrv_amd import capi
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    wl = sys.argv[2] if len(sys.argv) > 2 else "pq11_luv"
    src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, wl))
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    sha = capi.kernel_source_sha()
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
    except Exception:
        commit = "?"
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, wl)))
    for f in ("bench_plain.log", "bench_under_rocprof.log"):
        line = [l for l in open(os.path.join(src, f)).read().splitlines() if l.startswith('{"metric"')][-1]
        open(os.path.join(dst, "%s_%s_%s" % (tag, wl, f.replace(".log", ".json"))), "w").write(line + "\n")
    B, w, h = 20, 3840, 2160
    px = B * w * h
    c = {}
    for grp in ("fetch", "write", "inst", "wait", "mix1", "mix2"):
        p = os.path.join(src, "pmc_" + grp, "p_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for k, d in load(p).items():
            c.setdefault(k, {}).update(d)
    enc = next(k for k in c if "k_encode" in k)
    dec = next(k for k in c if "k_decode" in k)
    syn = next((k for k in c if "k_synth" in k), None)
    lines = ["# %s PMC summary -- workload %s, %d x %dx%d frames per launch (tools/prof_driver.py)" % (tag, wl, B, w, h), "",
             "rocprofv3 --pmc, one run per counter group; per-launch averages.  Kernel sources %s, commit %s." % (sha, commit), ""]
    alg = {"enc_r": 12.0 * px, "enc_w": 3.0 * px, "dec_r": 3.0 * px, "dec_w": 12.0 * px}
    mixes = {}
    for name, k, r_alg, w_alg in (("encode", enc, alg["enc_r"], alg["enc_w"]), ("decode", dec, alg["dec_r"], alg["dec_w"])):
        d = c[k]
        lines += ["## %s: `%s`" % (name, k), "", "| counter | per launch |", "|---|---|"]
        lines += ["| %s | %.4g |" % (cn, v) for cn, v in sorted(d.items())]
        fetch, write = d["FETCH_SIZE"] * 1024, d["WRITE_SIZE"] * 1024
        valu_px = d["SQ_INSTS_VALU"] * 64 / px
        lines += ["", "* FETCH_SIZE %.4g B raw (x2 = %.4g B) vs algorithmic read %.4g B; WRITE_SIZE %.4g B vs algorithmic write %.4g B"
                  % (fetch, 2 * fetch, r_alg, write, w_alg),
                  "* VALU instructions per pixel: %.1f; LDS instructions per pixel: %.2f; waves: %d"
                  % (valu_px, d["SQ_INSTS_LDS"] * 64 / px, d["SQ_WAVES"]),
                  "* wave time split: active %.0f %%, waiting on memory/LDS counters (SQ_WAIT_ANY) %.0f %%, issue-stalled %.0f %%"
                  % (100 * d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"],
                     100 * d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]),
                  "* LDS: %.0f %% of LDS-active cycles are bank-conflict cycles" % (100 * d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1))]
        if "SQ_INSTS_VALU_FMA_F32" in d:
            named = sum(d.get(cn, 0.0) for cn in COST)
            other = max(d["SQ_INSTS_VALU"] - named, 0.0)
            cyc = sum(d.get(cn, 0.0) * cost for cn, cost in COST.items()) + other * OTHER_COST
            fp64 = sum(d.get(cn, 0.0) for cn in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
            mixes[name] = {"issue_cycles_per_launch": cyc, "pixels_per_launch": float(px), "valu_per_pixel": round(valu_px, 1),
                           "fp64_per_pixel": round(fp64 * 64 / px, 1), "other_per_pixel": round(other * 64 / px, 1),
                           "trans_per_pixel": round(d.get("SQ_INSTS_VALU_TRANS_F32", 0.0) * 64 / px, 2)}
            lines += ["* VALU mix per pixel: fp32 add/mul/fma %.1f, int32 %.1f, fp64 %.1f, cvt %.1f, trans %.2f, other %.1f -> "
                      "%.0f issue cycles per pixel-wave (= %.2f cycles per instruction)"
                      % ((d.get("SQ_INSTS_VALU_ADD_F32", 0) + d.get("SQ_INSTS_VALU_MUL_F32", 0) + d.get("SQ_INSTS_VALU_FMA_F32", 0)) * 64 / px,
                         d.get("SQ_INSTS_VALU_INT32", 0) * 64 / px, fp64 * 64 / px, d.get("SQ_INSTS_VALU_CVT", 0) * 64 / px,
                         d.get("SQ_INSTS_VALU_TRANS_F32", 0) * 64 / px, other * 64 / px, cyc * 64 / px, cyc / max(d["SQ_INSTS_VALU"], 1))]
        lines += [""]
    if syn and "WRITE_SIZE" in c[syn]:
        lines += ["## calibration: `lh::k_synth` writes 3 x %d x %dx%d x 4 B = %.4g B with 4-B stores; WRITE_SIZE reports %.4g B"
                  % (3 * B, w, h, 3 * B * 3 * w * h * 4.0, c[syn]["WRITE_SIZE"] * 1024), ""]
    e = c[enc]
    traffic = 2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024
    lines += ["## roofline.traffic", "",
              "encode HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE = %.5g B; algorithmic = 15 B x %d px = %.5g B (ratio %.3f)"
              % (traffic, px, 15.0 * px, traffic / (15.0 * px)), ""]
    open(os.path.join(dst, "%s_%s_pmc_summary.md" % (tag, wl)), "w").write("\n".join(lines))
    update_json(os.path.join(dst, "traffic_latest.json"), wl,
                {"tag": tag, "workload": wl, "kernel_source_sha": sha, "commit": commit, "pixels_per_launch": float(px),
                 "hbm_bytes_per_launch": traffic, "fetch_size_bytes_raw": e["FETCH_SIZE"] * 1024,
                 "write_size_bytes": e["WRITE_SIZE"] * 1024,
                 "note": "2*FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC, separate passes (tools/profile_round.sh)"})
    if "encode" in mixes:
        update_json(os.path.join(dst, "valu_mix_latest.json"), wl,
                    dict(mixes["encode"], tag=tag, workload=wl, kernel_source_sha=sha, commit=commit,
                         note="PMC class counters x issue costs of tools/valu_bench.hip (tools/summarize_profile.py)"))
    print("\n".join(lines[-14:]))


if __name__ == "__main__":
    main()

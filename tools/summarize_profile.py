#!/usr/bin/env python3
"""tools/summarize_profile.py <tag> [workload] -- turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the
committed artefacts: profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_summary.md and
profiles/traffic_latest.json (per-launch HBM bytes that bench.py reports as roofline.traffic).

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE tallies the 128-B
requests of wide (16 B/lane) coalesced loads at 64 B (MI355X_MICROARCH.md, HBM section), hence the factor 2 on
the read side of the ENCODE kernel, whose pixel reads are all 16 B/lane.  Calibration inside the same run:
lh::k_synth writes a known byte count with 4-B stores (WRITE_SIZE ratio printed), and the decode kernel's
reads are 8-B / 4-B per lane (factor reported as measured/algorithmic, not assumed).
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    wl = sys.argv[2] if len(sys.argv) > 2 else "pq11_luv"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + (sys.argv[3] if len(sys.argv) > 3 else tag))
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, wl)))
    for f in ("bench_plain.log", "bench_under_rocprof.log"):
        line = [l for l in open(os.path.join(src, f)).read().splitlines() if l.startswith('{"metric"')][-1]
        open(os.path.join(dst, "%s_%s_%s" % (tag, wl, f.replace(".log", ".json"))), "w").write(line + "\n")
    B, w, h = 20, 3840, 2160
    px = B * w * h
    c = {}
    for grp in ("fetch", "write", "inst", "wait"):
        for k, d in load(os.path.join(src, "pmc_" + grp, "p_counter_collection.csv")).items():
            c.setdefault(k, {}).update(d)
    enc = next(k for k in c if "k_encode" in k)
    dec = next(k for k in c if "k_decode" in k)
    syn = next((k for k in c if "k_synth" in k), None)
    lines = ["# %s PMC summary -- workload %s, %d x %dx%d frames per launch (tools/prof_driver.py)" % (tag, wl, B, w, h), "",
             "rocprofv3 --pmc, one run per counter group; per-launch averages.", ""]
    alg = {"enc_r": 12.0 * px, "enc_w": 3.0 * px, "dec_r": 3.0 * px, "dec_w": 12.0 * px}
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
                  "* LDS: %.0f %% of LDS-active cycles are bank-conflict cycles" % (100 * d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1)), ""]
    if syn:
        known = 4.0 * c[syn].get("SQ_WAVES", 0)  # placeholder, real byte count below
        lines += ["## calibration: `lh::k_synth` writes 3 x %d x %dx%d x 4 B = %.4g B with 4-B stores; WRITE_SIZE reports %.4g B"
                  % (3 * B, w, h, 3 * B * 3 * w * h * 4.0, c[syn]["WRITE_SIZE"] * 1024), ""]
    e = c[enc]
    traffic = 2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024
    lines += ["## roofline.traffic", "",
              "encode HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE = %.5g B; algorithmic = 15 B x %d px = %.5g B (ratio %.3f)"
              % (traffic, px, 15.0 * px, traffic / (15.0 * px)), ""]
    open(os.path.join(dst, "%s_%s_pmc_summary.md" % (tag, wl)), "w").write("\n".join(lines))
    if wl == "pq11_luv":
        json.dump({"tag": tag, "workload": wl, "pixels_per_launch": float(px), "hbm_bytes_per_launch": traffic,
                   "fetch_size_bytes_raw": e["FETCH_SIZE"] * 1024, "write_size_bytes": e["WRITE_SIZE"] * 1024,
                   "note": "2*FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC, separate passes (tools/profile_round.sh)"},
                  open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
    print("\n".join(lines[-12:]))


if __name__ == "__main__":
    main()

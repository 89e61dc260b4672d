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
tools/bench/valu_bench.hip on MI355X (profiles/r02_valu_issue_rates.txt): fp32 add/mul/fma and int32 2 cycles, fp64 4,
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


def capture_sha(src):
    """the kernel-source SHA the capture RAN on: bench.py prints it in its JSON line (never the SHA of the local tree)"""
    line = [l for l in open(os.path.join(src, "bench_plain.log")).read().splitlines() if l.startswith('{"metric"')][-1]
    return json.loads(line).get("kernel_source_sha")


def launch_windows(trace_csv, pattern):
    """From a rocprofv3 kernel trace: the dispatches of one kernel grouped into back-to-back runs (gaps < 30 us) of >= 5
    launches.  Per run: whether its launches overlap each other (two lanes of an unordered section) and the run's span /
    number of launches -- the profiler-side counterpart of bench.py's `kernel_ms` (hipEvent window / K)."""
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(trace_csv))
                if pattern in r["Kernel_Name"])
    out = {"isolated_us": [], "ordered_us": [], "ordered_span_us": [], "overlapped_us": [], "overlapped_span_us": []}
    runs, cur = [], []
    for k in ks:
        if cur and k[0] - max(e for _, e in cur) >= 30000:
            runs.append(cur)
            cur = []
        cur.append(k)
    if cur:
        runs.append(cur)
    for run in runs:
        if len(run) < 5:
            out["isolated_us"] += [(e - b) / 1e3 for b, e in run]
            continue
        nov = sum(1 for i in range(1, len(run)) if run[i][0] < run[i - 1][1])
        kind = "overlapped" if nov > len(run) // 2 else "ordered"
        out[kind + "_us"] += [(e - b) / 1e3 for b, e in run]
        out[kind + "_span_us"].append((max(e for _, e in run) - run[0][0]) / 1e3 / len(run))
    return out


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float("nan")


def main():
    from lumahdrv_amd import capi
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if len(args) > 0 else "r03"
    wl = args[1] if len(args) > 1 else "pq11_luv"
    src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, wl))
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    sha = capture_sha(src)
    if sha != capi.kernel_source_sha():
        print("NOTE: the capture ran on kernel sources %s, the local tree is %s: bench.py will not report these figures "
              "until the tree matches again" % (sha, capi.kernel_source_sha()))
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
    except Exception:
        commit = "?"
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, wl)))
    if os.path.exists(os.path.join(src, "stats_ordered", "bench_kernel_stats.csv")):
        shutil.copy(os.path.join(src, "stats_ordered", "bench_kernel_stats.csv"),
                    os.path.join(dst, "%s_%s_kernel_stats_ordered.csv" % (tag, wl)))
    for f in ("bench_plain.log", "bench_under_rocprof.log", "bench_ordered_under_rocprof.log"):
        if not os.path.exists(os.path.join(src, f)):
            continue
        line = [l for l in open(os.path.join(src, f)).read().splitlines() if l.startswith('{"metric"')][-1]
        open(os.path.join(dst, "%s_%s_%s" % (tag, wl, f.replace(".log", ".json"))), "w").write(line + "\n")
    shape = json.load(open(os.path.join(src, "shape.json"))) if os.path.exists(os.path.join(src, "shape.json")) else {}
    B, w, h = shape.get("frames", 20), shape.get("width", 3840), shape.get("height", 2160)
    px = B * w * h
    c = {}
    for grp in ("fetch", "write", "inst", "wait", "mix1", "mix2"):
        p = os.path.join(src, "pmc_" + grp, "p_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for k, d in load(p).items():
            c.setdefault(k, {}).update(d)
    encs = sorted(k for k in c if "k_encode<" in k)
    # pq10_ycbcr: the half-input kernel (LM = 6: `<2, true, 4, 6, false>`) is the workload's encode kernel, the per-pixel one
    # (LM = 5) is what prof_driver.py ran on its float stream
    def lm(k):
        return k.split("k_encode<")[1].split(">")[0].replace(" ", "").split(",")[3]
    enc = next((k for k in encs if lm(k) == "6"), encs[0])
    enc_float = next((k for k in encs if lm(k) == "5" and k != enc), None)
    dec = next(k for k in c if "k_decode<" in k)
    syn = next((k for k in c if "k_synth" in k), None)
    eprobe = next((k for k in c if "k_encode_traffic_probe" in k), None)
    dprobe = next((k for k in c if "k_decode_traffic_probe" in k), None)
    lines = ["# %s PMC summary -- workload %s, %d x %dx%d frames per launch (tools/prof_driver.py)" % (tag, wl, B, w, h), "",
             "rocprofv3 --pmc, one run per counter group; per-launch averages.  Kernel sources %s (the SHA the capture ran on), "
             "summarised at commit %s." % (sha, commit), ""]
    alg = {"enc_r": 12.0 * px, "enc_w": 3.0 * px, "dec_r": 3.0 * px, "dec_w": 12.0 * px}
    # calibration on the traffic-only probes: same access pattern, known byte counts
    cal = {"enc_r": 2.0, "enc_w": 1.0, "dec_r": 2.0, "dec_w": 1.0}
    cal_src = {k: "guide (gfx950: FETCH_SIZE x 2 for wide loads; WRITE_SIZE as reported)" for k in cal}
    for probe, r_key, w_key in ((eprobe, "enc_r", "enc_w"), (dprobe, "dec_r", "dec_w")):
        if probe and "FETCH_SIZE" in c[probe] and "WRITE_SIZE" in c[probe]:
            cal[r_key] = alg[r_key] / (c[probe]["FETCH_SIZE"] * 1024)
            cal[w_key] = alg[w_key] / (c[probe]["WRITE_SIZE"] * 1024)
            cal_src[r_key] = cal_src[w_key] = "measured on `%s` in the same run" % probe
    mixes, traffic = {}, {}
    kernels = [("encode", enc, "enc_r", "enc_w"), ("decode", dec, "dec_r", "dec_w")]
    if enc_float:
        kernels.append(("encode_float", enc_float, "enc_r", "enc_w"))
    for name, k, r_key, w_key in kernels:
        d = c[k]
        lines += ["## %s: `%s`" % (name, k), "", "| counter | per launch |", "|---|---|"]
        lines += ["| %s | %.4g |" % (cn, v) for cn, v in sorted(d.items())]
        fetch, write = d["FETCH_SIZE"] * 1024, d["WRITE_SIZE"] * 1024
        traffic[name] = cal[r_key] * fetch + cal[w_key] * write
        valu_px = d["SQ_INSTS_VALU"] * 64 / px
        lines += ["", "* FETCH_SIZE %.4g B raw, x %.4f (%s) = %.4g B vs algorithmic read %.4g B; WRITE_SIZE %.4g B x %.4f = %.4g B vs "
                  "algorithmic write %.4g B" % (fetch, cal[r_key], cal_src[r_key], cal[r_key] * fetch, alg[r_key], write, cal[w_key],
                                              cal[w_key] * write, alg[w_key]),
                  "* HBM traffic per launch = %.5g B = %.4f x algorithmic (15 B x %d px)" % (traffic[name], traffic[name] / (15.0 * px), px),
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
    lines += ["## calibration", ""]
    if syn and "WRITE_SIZE" in c[syn]:
        lines += ["* `lh::k_synth` (4-B stores): WRITE_SIZE reports %.5g B per launch" % (c[syn]["WRITE_SIZE"] * 1024)]
    for probe, what in ((eprobe, "12 B read (16 B/lane loads) + 3 B written per pixel"), (dprobe, "3 B read (8 / 4 B per lane loads) + 12 B written per pixel")):
        if probe and "FETCH_SIZE" in c[probe]:
            lines += ["* `%s` (%s, known byte counts): FETCH_SIZE %.5g B, WRITE_SIZE %.5g B" % (probe, what, c[probe]["FETCH_SIZE"] * 1024, c[probe].get("WRITE_SIZE", 0) * 1024)]
    lines += [""]
    # ---- kernel durations under the profiler against bench.py's own clock
    tr_csv = os.path.join(src, "stats", "bench_kernel_trace.csv")
    if os.path.exists(tr_csv):
        lines += ["## kernel durations in the rocprofv3 kernel trace of `python bench.py` (the same command, under the profiler)", "",
                  "bench.py times a region of K launches with a hipEvent pair and reports window / K (`kernel_ms`).  By default the K",
                  "launches of a region run in an unordered section (two lanes), so two launches are in flight and each one's OWN",
                  "duration in the trace is about twice the per-launch window; the `*_ordered` legs of the same run are K launches back",
                  "to back on one stream, whose trace durations are directly comparable with `kernel_ms_ordered`.", "",
                  "| kernel | launches | ordered: median duration | ordered: span / launches | overlapped: median duration | overlapped: span / launches | isolated launches: median |",
                  "|---|---|---|---|---|---|---|"]
        for name, pat in (("encode", "k_encode<"), ("decode", "k_decode<")):
            wv = launch_windows(tr_csv, pat)
            lines += ["| %s | %d | %.1f us | %.1f us | %.1f us | %.1f us | %.1f us |"
                      % (name, sum(len(wv[k]) for k in ("isolated_us", "ordered_us", "overlapped_us")), med(wv["ordered_us"]),
                         med(wv["ordered_span_us"]), med(wv["overlapped_us"]), med(wv["overlapped_span_us"]), med(wv["isolated_us"]))]
        for f, label in (("bench_under_rocprof.log", "under rocprofv3"), ("bench_plain.log", "same box, no profiler")):
            try:
                line = [l for l in open(os.path.join(src, f)).read().splitlines() if l.startswith('{"metric"')][-1]
                j = json.loads(line)
                rf, drf = j["roofline"], j.get("decode_roofline", {})
                drf = drf.get("hbm", drf)
                lines += ["", "bench.py %s: encode kernel_ms %.4f (ordered %s, isolated %s), decode kernel_ms %s (ordered %s)"
                          % (label, rf["kernel_ms"], rf.get("kernel_ms_ordered"), rf.get("kernel_ms_isolated_launch"),
                             drf.get("kernel_ms"), drf.get("kernel_ms_ordered"))]
            except Exception:
                pass
        so = os.path.join(src, "stats_ordered", "bench_kernel_stats.csv")
        if os.path.exists(so):
            lines += ["", "`rocprofv3 --stats` of the same command with `--lanes 0` (`%s_%s_kernel_stats_ordered.csv`: every launch alone on" % (tag, wl),
                      "its stream, so the profiler's average IS the per-launch time):", ""]
            for r in csv.DictReader(open(so)):
                if "k_encode<" in r["Name"] or "k_decode<" in r["Name"]:
                    lines += ["* `%s`: %s calls, average %.1f us (min %.1f, max %.1f)"
                              % (r["Name"].split("(")[0].replace("void ", ""), r["Calls"], float(r["AverageNs"]) / 1e3,
                                 float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)]
            try:
                line = [l for l in open(os.path.join(src, "bench_ordered_under_rocprof.log")).read().splitlines() if l.startswith('{"metric"')][-1]
                j = json.loads(line)
                drf = j.get("decode_roofline", {})
                drf = drf.get("hbm", drf)
                lines += ["* bench.py's own hipEvent figures in that run: encode kernel_ms %.4f, decode kernel_ms %s"
                          % (j["roofline"]["kernel_ms"], drf.get("kernel_ms"))]
            except Exception:
                pass
        lines += ["", "(In the default capture the trace's span per launch of the overlapped runs is not the hipEvent window of the same run:",
                  "start timestamps of concurrently queued dispatches are taken when the packet is processed.  The ordered figures agree",
                  "with bench.py's to a fraction of a per cent and are the ones to cross-check `kernel_ms_ordered` with.)", ""]
    lines += ["## roofline.traffic", "",
              "encode HBM bytes per launch = %.5g B; algorithmic = 15 B x %d px = %.5g B (ratio %.4f)"
              % (traffic["encode"], px, 15.0 * px, traffic["encode"] / (15.0 * px)),
              "decode HBM bytes per launch = %.5g B (ratio %.4f)" % (traffic["decode"], traffic["decode"] / (15.0 * px)), ""]
    open(os.path.join(dst, "%s_%s_pmc_summary.md" % (tag, wl)), "w").write("\n".join(lines))
    e, dd = c[enc], c[dec]
    update_json(os.path.join(dst, "traffic_latest.json"), wl,
                {"tag": tag, "workload": wl, "kernel_source_sha": sha, "commit": commit, "pixels_per_launch": float(px),
                 "hbm_bytes_per_launch": traffic["encode"], "decode_hbm_bytes_per_launch": traffic["decode"],
                 "fetch_size_bytes_raw": e["FETCH_SIZE"] * 1024, "write_size_bytes": e["WRITE_SIZE"] * 1024,
                 "decode_fetch_size_bytes_raw": dd["FETCH_SIZE"] * 1024, "decode_write_size_bytes": dd["WRITE_SIZE"] * 1024,
                 "calibration": {k: round(v, 5) for k, v in cal.items()},
                 "note": "cal_r * FETCH_SIZE + cal_w * WRITE_SIZE, rocprofv3 PMC, separate passes (tools/profile_round.sh); calibration "
                         "factors measured on the traffic-only probes of the same access patterns in the same run (guide: x2 for "
                         "wide loads on gfx950)"})
    if "encode" in mixes:
        update_json(os.path.join(dst, "valu_mix_latest.json"), wl,
                    dict(mixes["encode"], tag=tag, workload=wl, kernel_source_sha=sha, commit=commit, decode=mixes.get("decode"),
                         encode_float=mixes.get("encode_float"),
                         note="PMC class counters x issue costs of tools/bench/valu_bench.hip (tools/summarize_profile.py)"))
    print("\n".join(lines[-24:]))


if __name__ == "__main__":
    main()

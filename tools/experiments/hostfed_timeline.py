#!/usr/bin/env python3
"""tools/experiments/hostfed_timeline.py [frame] -- from the rocprofv3 traces of tools/experiments/hostfed_trace.sh (gpurun_out/prof_hf2160/): the copies,
kernels and long HIP calls of ONE LumaEncoder::encode(LumaFrame*) call on a pageable 3840x2160 frame, and the per-frame
period -> profiles/r03_hostfed_timeline.txt"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = os.path.join(ROOT, "gpurun_out", "prof_hf2160")


def rows(name):
    return list(csv.DictReader(open(os.path.join(D, name))))


def main():
    f = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    api, mc, kt = rows("hf_hip_api_trace.csv"), rows("hf_memory_copy_trace.csv"), rows("hf_kernel_trace.csv")
    enc = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in kt if "k_encode<" in r["Kernel_Name"])
    nb = 4                                            # bands per frame in the traced build
    ks = enc[nb * f:nb * f + nb]
    h2d = sorted(int(r["Start_Timestamp"]) for r in mc if r["Direction"].endswith("HOST_TO_DEVICE"))
    # the frame's first upload chunk: the first H2D copy after the previous frame's last kernel
    prev_last = enc[nb * f - 1][1]
    a = min(t for t in h2d if t > prev_last - 2600000 and t > enc[nb * f - 1][0])
    b = min(t for t in h2d if t > ks[-1][1])
    print("# one LumaEncoder::encode(LumaFrame*) call, pageable 3840x2160 frame (99.5 MB up, 24.9 MB down), frame %d of the run;" % f)
    print("# times in us from the call's first upload chunk; period to the next call's first chunk: %.0f us = %.0f Mpixel/s"
          % ((b - a) / 1e3, 3840 * 2160 / ((b - a) / 1e9) / 1e6))
    ev = []
    for r in api:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if a <= s < b and e - s > 20000:
            ev.append((s, "host  " + r["Function"], e - s))
    for r in mc:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if a <= s < b:
            ev.append((s, "copy  " + ("H2D" if r["Direction"].endswith("HOST_TO_DEVICE") else "D2H"), e - s))
    for s, e in ks:
        ev.append((s, "kernel k_encode (one row band)", e - s))
    for s, n, d in sorted(ev):
        print("%9.1f  %-34s %7.1f us" % ((s - a) / 1e3, n, d / 1e3))
    last_up = max(s + d for s, n, d in ev if n.endswith("H2D"))
    print("# upload phase (first H2D start .. last H2D end): %.0f us; after it until the next call's first chunk: %.0f us"
          % ((last_up - a) / 1e3, (b - last_up) / 1e3))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/bench/launch_rules_sweep.py [--quick] -- are the launch-geometry rules of grid_for / block_threads_for
(lumahdrv_amd/csrc/lumahip_launch.hip) right away from the shape they were found on (20 x 3840x2160)?

For {1280x720, 1920x1080, 3840x2160, 7680x4320} x {1, 2, 4, 8, 20, 50 frames per launch} x {PQ-11 Lu'v', HDR10 YCbCr} x
{encode, decode}: the median kernel time of isolated launches (hipEvent pair around ONE launch, distinct device-resident
batches totalling >= 1 GB so that nothing is served from L2 / the Infinity Cache) with the rule (blocks_per_cu = 0) and with
every fixed number of persistent workgroups per CU in CANDIDATES.  One line per shape: the rule's time, the best fixed
setting and its time, and the rule's loss against it; a summary of the worst losses at the end.  -> profiles/r03_launch_rules.txt

All settings of one shape are measured interleaved in one process on the same buffers (round-robin over the settings inside
every repetition), so box-to-box and placement differences cancel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402

CANDIDATES = (2, 3, 4, 5, 6, 8, 12, 18)
WORKLOADS = {"pq11_luv": (L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, 1.0), "pq10_ycbcr": (L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01, 20.0)}
SIZES = ((1280, 720), (1920, 1080), (3840, 2160), (7680, 4320))
BATCHES = (1, 2, 4, 8, 20, 50)


def main():
    quick = "--quick" in sys.argv
    # --profile N: the VP9 profile (plane layout) swept, default 2 = 4:2:0 16-bit; --small: 1080p and 4K, 1 / 4 / 20 frames only
    profile = int(sys.argv[sys.argv.index("--profile") + 1]) if "--profile" in sys.argv else 2
    sizes = SIZES[1:3] if "--small" in sys.argv else SIZES
    batches = (1, 4, 20) if "--small" in sys.argv else BATCHES
    print("VP9 profile %d" % profile)
    dev = torch.device("cuda:0")
    worst = []
    for wl, (ptf, bits, cs, bitsC, mx, mn, sc) in WORKLOADS.items():
        lut = L.build_lut(ptf, bits, mx, mn)
        ctxs = {}
        for pcu in (0,) + CANDIDATES:
            c = L.Context(0)
            c.tune("blocks_per_cu", pcu)
            c.set_stream(torch.cuda.current_stream().cuda_stream)
            c.set_quantizer(ptf, bits, cs, bitsC, mx, mn, lut)
            ctxs[pcu] = c
        for direction in (0, 1):
            print("== %s %s: median us per launch; rule | best fixed workgroups-per-CU | rule's loss ==" % (wl, "encode" if direction == 0 else "decode"),
                  flush=True)
            for (w, h) in sizes:
                for B in batches:
                    n3 = 3 * w * h
                    _, hs, st, _ = L.plane_geometry(w, h, profile)
                    psz = [hs[p] * st[p] for p in range(3)]
                    batch_bytes = B * (n3 * 4 + sum(psz))
                    if batch_bytes > 24e9:
                        continue
                    nb = int(max(3, min(48, -(-1.2e9 // batch_bytes))))
                    src = torch.empty(nb * B * n3, dtype=torch.float32, device=dev)
                    planes = [torch.zeros(nb * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
                    ctxs[0].synth_frames_device(src.data_ptr(), n3, nb * B, w, h)
                    for b in range(nb):      # planes hold real codes for the decode side
                        ctxs[0].encode_frames_device(src.data_ptr() + b * B * n3 * 4, n3, B, w, h, sc, profile,
                                                     [planes[p].data_ptr() + b * B * psz[p] for p in range(3)], st, psz)
                    torch.cuda.synchronize()
                    reps = 2 if quick else max(2, min(6, 72 // nb))
                    ms = {k: [] for k in ctxs}
                    i = 0
                    for rep in range(reps + 1):
                        for b in range(nb):
                            for k, c in ctxs.items():
                                bb = (b + i) % nb
                                i += 1
                                t = c.time_launches(direction, 1, src.data_ptr() + bb * B * n3 * 4, n3, B, w, h, sc, profile,
                                                    [planes[p].data_ptr() + bb * B * psz[p] for p in range(3)], st, psz)
                                if rep > 0:        # first round = warm-up
                                    ms[k].append(t)
                    med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
                    best = min(CANDIDATES, key=lambda k: med[k])
                    loss = med[0] / med[best] - 1.0
                    worst.append((loss, wl, direction, w, h, B, best))
                    print("  %4dx%-4d x%-2d  rule %8.1f (%5.1f Gpx/s) | best %2d/CU %8.1f | %+5.1f %%   [%s]"
                          % (w, h, B, med[0] * 1e3, B * w * h / med[0] / 1e6, best, med[best] * 1e3, 100 * loss,
                             " ".join("%d:%.1f" % (k, med[k] * 1e3) for k in CANDIDATES)), flush=True)
                    del src, planes
        for c in ctxs.values():
            c.close()
    worst.sort(reverse=True)
    print("== shapes where the rule loses most against the best fixed setting ==")
    for loss, wl, d, w, h, B, best in worst[:12]:
        print("  %+5.1f %%  %s %s %dx%d x%d (best %d/CU)" % (100 * loss, wl, "encode" if d == 0 else "decode", w, h, B, best))


if __name__ == "__main__":
    main()

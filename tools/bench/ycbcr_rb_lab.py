#!/usr/bin/env python3
"""tools/bench/ycbcr_rb_lab.py [only] -- YCbCr decode (HDR10 recipe, 20 x 3840x2160 per launch) with red and blue read from the
per-stream (Y', Cr) -> R / (Y', Cb) -> B tables in global memory (lumahip_tune ycbcr_rb_tables 1: k_decode<CS_YCBCR, ., ., ., ., YT, RB>,
two 4-byte gathers + two powf per pixel) against six powf per pixel (ycbcr_rb_tables 0), on two kinds of content:
  random    the synthetic stream of SURVEY 8(d): every pixel independent, so consecutive lanes gather from unrelated table rows;
  coherent  the same stream low-pass filtered in the log domain (32 x 32 box, bilinear up) -- neighbouring pixels have
            neighbouring codes, as in video.
plus smooth pictures with per-pixel noise and finely textured ones (make_content).  Variants: six_powf (ycbcr_rb_tables 0),
always (2: every wave gathers), adaptive (1, the default: a wave gathers for a unit when its lanes' codes are close,
luma_kernels.hpp rb_wave_near); `--sweep` tries a range of closeness bounds.  All contexts decode the same planes interleaved in
one process; outputs must be bit-equal.  `<variant>_<content>` as first argument runs that pair alone (for rocprofv3 --pmc).
-> profiles/r05_ycbcr_decode_tables.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def make_content(src, kind, nb, B, h, w):
    """in place on the synthetic stream in `src`: "random" leaves it; "coherent" low-pass filters it in the log domain (32 x 32 box,
    bilinear up); "noisy<s>" = coherent times exp(N(0, s/100)) per pixel and channel (sensor noise / film grain on a smooth picture);
    "detail" = a 4 x 4 box instead of 32 x 32 (fine texture everywhere)"""
    if kind == "random":
        return
    if kind.startswith("narrow"):
        # every pixel independent again, but from a narrow part of the gamut: one grey level per pixel, log-uniform over <n> stops
        # around 10 cd/m2 (before preScaling), each channel off it by up to +-<n> % -- no spatial coherence at all, yet only a
        # small part of the red / blue tables is ever read
        stops = int(kind[6:])
        g = torch.Generator(device=src.device)
        g.manual_seed(9)
        v = src.view(nb * B, 3, h * w)
        for f in range(v.shape[0]):
            base = torch.exp2((torch.rand(h * w, device=src.device, generator=g) - 0.5) * stops) * 0.5
            v[f] = base[None] * (1.0 + (torch.rand(3, h * w, device=src.device, generator=g) - 0.5) * (0.02 * stops))
        torch.cuda.synchronize()
        return
    box = 4 if kind == "detail" else 32
    g = torch.Generator(device=src.device)
    g.manual_seed(5)
    v = src.view(nb * B * 3, 1, h, w)
    for i in range(v.shape[0]):
        lg = torch.log(v[i:i + 1])
        lo = F.interpolate(F.avg_pool2d(lg, box), size=(h, w), mode="bilinear", align_corners=False)
        if kind.startswith("noisy"):
            lo += torch.randn(lo.shape, device=src.device, generator=g) * (int(kind[5:]) / 100.0)
        v[i:i + 1] = torch.exp(lo)
    torch.cuda.synchronize()


def main():
    args = [a for a in sys.argv[1:]]
    only = args[0] if args and not args[0].startswith("--") else ""
    sweep = "--sweep" in args
    dev = torch.device("cuda:0")
    bits = bitsC = int(os.environ.get("RB_LAB_BITS", "10"))        # RB_LAB_BITS=12: 2 x 64 MiB of tables
    ptf, cs, mx, mn, sc = L.PTF_PQ, L.CS_YCBCR, 1000.0, 0.01, 20.0
    lut = L.build_lut(ptf, bits, mx, mn)
    ctxs = {}
    variants = {"six_powf": (0, None), "adaptive": (1, None), "always": (2, None)}
    if "--vw2" in args:   # two pixels per thread and row (75 VGPRs -> 6 waves per SIMD instead of 4)
        variants = {"six_powf": (0, None), "six_vw2": (0, "vw2"), "always": (2, None), "always_vw2": (2, "vw2"), "adaptive_vw2": (1, "vw2")}
    if sweep:     # closeness bounds of the adaptive mode (luma_kernels.hpp rb_wave_local): luminance codes, colour codes
        variants = {"six_powf": (0, None), "always": (2, None)}
        for ny, nc in ((16, 8), (32, 12), (64, 24), (128, 48), (256, 96), (512, 200)):
            variants["y%d_c%d" % (ny, nc)] = (1, (ny, nc))
    for key, (mode, near) in variants.items():
        c = L.Context(0)
        c.tune("ycbcr_rb_tables", mode)
        if near == "vw2":
            c.tune("dec_vw", 2)
        elif near:
            c.tune("rb_near_y", near[0])
            c.tune("rb_near_c", near[1])
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.set_quantizer(ptf, bits, cs, bitsC, mx, mn, lut)
        ctxs[key] = c
    w, h, B, nb = 3840, 2160, 20, 3
    n1, n3 = w * h, 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    psz = [hs[p] * st[p] for p in range(3)]
    src = torch.empty(nb * B * n3, dtype=torch.float32, device=dev)
    planes = [torch.zeros(nb * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    ref = torch.empty(nb * B * n3, dtype=torch.float32, device=dev)
    out = torch.empty(nb * B * n3, dtype=torch.float32, device=dev)
    print("us per launch (20 x 3840x2160, HDR10 recipe decode), median; every variant's output bit-equal to six_powf's: checked")
    print("content  | " + " | ".join("%9s" % k for k in ctxs))
    for content in ("random", "coherent", "noisy2", "noisy5", "noisy10", "noisy20", "detail", "narrow3", "narrow8", "narrow16"):
        if only and not only.endswith("_" + content):
            continue
        ctxs["six_powf"].synth_frames_device(src.data_ptr(), n3, nb * B, w, h)
        torch.cuda.synchronize()
        make_content(src, content, nb, B, h, w)
        for b in range(nb):
            ctxs["six_powf"].encode_frames_device(src.data_ptr() + b * B * n3 * 4, n3, B, w, h, sc, 2,
                                                  [planes[p].data_ptr() + b * B * psz[p] for p in range(3)], st, psz)
        torch.cuda.synchronize()
        for k, c in ctxs.items():          # a new stream: the launch-level policy of mode 1 starts afresh
            if variants[k][0] == 1:
                c.tune("ycbcr_rb_tables", 1)
        ms = {k: [] for k in ctxs}
        for rep in range(1 if only else 3):
            for b in range(nb):
                for k, c in ctxs.items():
                    if only and not only.startswith(k):
                        continue
                    o = ref if k == "six_powf" else out
                    t = c.time_launches(1, 1, o.data_ptr() + b * B * n3 * 4, n3, B, w, h, sc, 2,
                                        [planes[p].data_ptr() + b * B * psz[p] for p in range(3)], st, psz)
                    if rep > 0 or only:
                        ms[k].append(t)
                    if k != "six_powf" and not only and rep == 0:
                        sl = slice(b * B * n3, (b + 1) * B * n3)
                        assert torch.equal(out[sl].view(torch.int32), ref[sl].view(torch.int32)), (content, k, b)
        torch.cuda.synchronize()
        med = {k: (sorted(v)[len(v) // 2] if v else float("nan")) for k, v in ms.items()}
        print("%-8s | " % content + " | ".join("%9.1f" % (1e3 * med[k]) for k in ctxs), flush=True)


if __name__ == "__main__":
    main()

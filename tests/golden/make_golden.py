#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libluma_ref.so, i.e.
/root/reference/src/luma_quantizer.cpp compiled unmodified by oracle/Makefile).

Runs only in the build container.  The fixtures are data (inputs + the reference's outputs); they are
what pins the oracle and the HIP path on the GPU box, where /root/reference does not exist.

Environment the values are tied to: glibc 2.35 libm (powf / log10f are called by the reference for the
LUTs and for the per-pixel YCbCr transform), x86-64, g++ -O2 -ffp-contract=off.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as o  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# name -> (ptf, bits, cs, bitsC, maxLum, minLum)
CONFIGS = {
    "pq11_luv8": (o.PTF_PQ, 11, o.CS_LUV, 8, 1e4, 0.005),          # C1 / C2 / C5 (test_simple_enc defaults)
    "pq10_ycbcr10": (o.PTF_PQ, 10, o.CS_YCBCR, 10, 1000.0, 0.01),   # C3 (HDR10 recipe)
    "log12_luv8": (o.PTF_LOG, 12, o.CS_LUV, 8, 1e4, 0.005),         # C4
    "linear12_xyz": (o.PTF_LINEAR, 12, o.CS_XYZ, 8, 1e4, 0.005),
    "pq8_luv8": (o.PTF_PQ, 8, o.CS_LUV, 8, 1e4, 0.005),             # 8-bit profiles
    "pq12_rgb": (o.PTF_PQ, 12, o.CS_RGB, 8, 1e4, 0.005),
    "psi11_luv8": (o.PTF_PSI, 11, o.CS_LUV, 8, 1e4, 0.005),         # decoder default PTF
    "hdrvdp12_luv10": (o.PTF_JND_HDRVDP, 12, o.CS_LUV, 10, 1e4, 0.005),
}


def special_frame(h=8, w=16, seed=7):
    """(3,h,w) float32: log-uniform positives with the edge cases the reference's arithmetic can meet"""
    rng = np.random.default_rng(seed)
    f = np.exp(rng.uniform(np.log(1e-4), np.log(2e4), size=(3, h, w))).astype(np.float32)
    sp = [(0, 0, 0), (1e-6, 1e-6, 1e-6), (1, 1, 1), (100, 100, 100), (10000, 0, 0), (0, 10000, 0), (0, 0, 10000),
          (0.5, 20, 3), (-5, 2, 1), (np.nan, 1, 1), (np.inf, 1, 1), (65504, 65504, 65504), (1e9, 1e9, 1e9),
          (-np.inf, 3, 3), (1e-30, 1e-30, 1e-30), (3e38, 3e38, 3e38)]
    for i, (r, g, b) in enumerate(sp):
        f[:, 0, i] = (r, g, b)
    # a constant-colour 2x2 quad block so that chroma averaging is exercised on equal values
    f[:, 2:4, 0:2] = np.array([0.5, 20, 3], dtype=np.float32)[:, None, None]
    return f


def extreme_frame(h=4, w=32):
    """(3,h,w) float32 of inputs at the edges of fp32: inf - inf patterns inside the RGB -> XYZ rows, +-FLT_MAX (partial sums
    that overflow), NaNs of both signs in every position, -0, denormals.  Not a stored fixture: both sides of a comparison
    evaluate it (oracle against the live reference on CPU, HIP against the oracle on the GPU)."""
    M = np.finfo(np.float32).max
    nan_neg = np.array([0xffc00000], dtype=np.uint32).view(np.float32)[0]
    tiny = np.float32(1e-45)
    px = [(np.inf, -np.inf, 1), (np.inf, 1, -np.inf), (1, np.inf, -np.inf), (-np.inf, np.inf, np.inf), (np.inf, np.inf, np.inf),
          (-np.inf, -np.inf, -np.inf), (M, M, M), (-M, M, M), (M, -M, M), (M, M, -M), (-M, -M, -M), (M, 0, 0), (0, M, 0), (0, 0, M),
          (np.nan, np.nan, np.nan), (1, np.nan, 1), (1, 1, np.nan), (nan_neg, 1, 1), (1, nan_neg, 1), (1, 1, nan_neg),
          (-0.0, -0.0, -0.0), (-0.0, 1, 1), (tiny, tiny, tiny), (tiny, 1, 1), (1e-38, 1e-38, 1e-38), (1e38, 1e-38, 1),
          (3.4e38, 1e-4, 1e-4), (1e-4, 3.4e38, 1e-4), (1e-4, 1e-4, 3.4e38), (-1e-45, 2, 2), (5e4, 5e4, 5e4), (0, 0, 0)]
    f = np.ones((3, h, w), dtype=np.float32)
    for i, p in enumerate(px[:w]):
        f[:, 0, i] = np.array(p, dtype=np.float32)
        f[:, 2, i] = np.array(p, dtype=np.float32)       # a second copy on another quad row (different averaging partners)
    f[:, 3, :] = np.float32(0.25)
    return f


def main():
    o.build(ref=True)
    assert o.have_ref()
    luts, tr, qv = {}, {}, {}
    frame = special_frame()
    tr["input"] = frame
    rng = np.random.default_rng(11)
    for name, cfg in CONFIGS.items():
        r = o.RefQuantizer(*cfg)
        m = r.mapping
        luts[name] = m
        # quantize: values around every LUT entry (exact hits, half-way points, neighbours) + wide range
        mids = (m[:-1].astype(np.float64) + m[1:].astype(np.float64)) / 2
        vals = np.concatenate([m, np.nextafter(m, np.float32(np.inf)), np.nextafter(m, np.float32(-np.inf)),
                               mids.astype(np.float32),
                               np.nextafter(mids.astype(np.float32), np.float32(np.inf)),
                               np.exp(rng.uniform(np.log(1e-6), np.log(1e6), 4096)).astype(np.float32),
                               np.array([0, -1, -0.0, np.nan, np.inf, -np.inf, 1e-45, 3e38], dtype=np.float32)])
        if m.size > 1024:  # keep the fixture small: subsample the dense part deterministically
            keep = rng.choice(vals.size - 8, 6000, replace=False)
            vals = np.concatenate([vals[np.sort(keep)], vals[-8:]])
        qv[name + "_in0"] = vals
        qv[name + "_q0"] = r.quantize_array(vals, 0).astype(np.uint16)
        cvals = np.concatenate([rng.uniform(-0.2, 1.2, 2048).astype(np.float32),
                                (np.arange(0, 2 * (2 ** cfg[3] - 1) + 1, dtype=np.float32) / (2 * (2 ** cfg[3] - 1))),
                                np.array([np.nan, np.inf, -np.inf, 0.0, 1.0], dtype=np.float32)])
        qv[name + "_in1"] = cvals
        qv[name + "_q1"] = r.quantize_array(cvals, 1).astype(np.uint16)
        codes = np.arange(-2, 2 ** cfg[1] + 2, dtype=np.float32)
        qv[name + "_dq0"] = r.dequantize_array(codes, 0)
        ccodes = np.arange(0, 2 ** cfg[3], dtype=np.float32)
        qv[name + "_dq1"] = r.dequantize_array(ccodes, 1)
        # colour transform, both directions, two scalings
        for sc in (1.0, 20.0, 0.25):
            f = frame.copy()
            r.transform(f, True, sc)
            tr["%s_fwd_sc%g" % (name, sc)] = f
            # decode direction input: what getVpxChannels would produce = dequantized codes
            h, w = frame.shape[1:]
            c0 = m[rng.integers(0, m.size, size=(h, w))]
            if cfg[2] in (o.CS_RGB, o.CS_XYZ):
                c1 = m[rng.integers(0, m.size, size=(h, w))]
                c2 = m[rng.integers(0, m.size, size=(h, w))]
            else:
                maxc = 2 ** cfg[3] - 1
                c1 = r.dequantize_array(rng.integers(0, maxc + 1, size=h * w).astype(np.float32), 1).reshape(h, w)
                c2 = r.dequantize_array(rng.integers(0, maxc + 1, size=h * w).astype(np.float32), 2).reshape(h, w)
            g = np.stack([c0, c1, c2]).astype(np.float32)
            tr["%s_inv_in_sc%g" % (name, sc)] = g.copy()
            r.transform(g, False, sc)
            tr["%s_inv_sc%g" % (name, sc)] = g
    # ---- plane loops: LumaEncoder::setChannels / LumaDecoder::getVpxChannels, the reference's own code
    # (oracle/_ref/ref_planes_tool), every profile, both directions, ragged sizes and a decoder-chosen stride
    assert o.have_ref_planes()
    pl = {}
    for name in ("pq11_luv8", "pq10_ycbcr10", "pq12_rgb", "pq8_luv8"):
        cfg = CONFIGS[name]
        rp = o.RefPlanes(*cfg)
        for (w, h) in ((34, 18), (64, 32)):
            f = o.synth_frame(w, h, frame=5)
            f[:, 0, :4] = [[np.nan, 0, -1, 7e4], [1, 0, 5, 7e4], [1, 0, 2, 7e4]]
            for profile in (0, 1, 2, 3):
                if (cfg[1] > 8) != (profile > 1):
                    continue                    # 8-bit tables with 8-bit samples, deeper tables with 16-bit samples
                key = "%s_%dx%d_p%d" % (name, w, h, profile)
                planes, st, mean = rp.encode(f, 1.0 if cfg[2] != o.CS_YCBCR else 20.0, profile)
                pl[key + "_in"] = f
                for p in range(3):
                    pl[key + "_plane%d" % p] = planes[p]
                pl[key + "_stride"] = np.array(st, dtype=np.int32)
                pl[key + "_mean"] = np.array([np.nan if mean is None else mean], dtype=np.float64)
                # decode direction: the encoded planes with a few out-of-range / garbage codes, odd strides
                dst = tuple(int(s) + 6 for s in st)
                dplanes = []
                for p in range(3):
                    q = np.full((planes[p].shape[0], dst[p]), 0x5A, dtype=np.uint8)
                    q[:, :st[p]] = planes[p]
                    q[0, :8] = [0xFF, 0xFF, 0x00, 0x00, 0x34, 0x12, 0xFF, 0x7F]
                    dplanes.append(q)
                sc = 1.0 if cfg[2] != o.CS_YCBCR else 20.0
                pl[key + "_dec_stride"] = np.array(dst, dtype=np.int32)
                for p in range(3):
                    pl[key + "_dec_plane%d" % p] = dplanes[p]
                pl[key + "_unpacked"] = rp.decode(dplanes, dst, w, h, sc, profile, xform=False)
                pl[key + "_decoded"] = rp.decode(dplanes, dst, w, h, sc, profile, xform=True)
    # decode digests the survey lacks: testFrame 1280x720 through the reference's encode -> decode, profile 2 and 3
    rp = o.RefPlanes(*CONFIGS["pq11_luv8"])
    dig = {}
    for profile in (2, 3):
        planes, st, mean = rp.encode(o.test_frame(1280, 720), 1.0, profile)
        dec = rp.decode(planes, st, 1280, 720, 1.0, profile)
        dig["testframe_1280x720_p%d" % profile] = {
            "Y": o.survey_digest(o.packed_rows(planes[0], 2560)), "U": o.survey_digest(o.packed_rows(planes[1], 2560 if profile == 3 else 1280)),
            "V": o.survey_digest(o.packed_rows(planes[2], 2560 if profile == 3 else 1280)),
            "decoded": o.survey_digest(dec), "mean_luminance_printed": mean}
    import json
    json.dump(dig, open(os.path.join(OUT, "ref_plane_digests.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "ref_planes.npz"), **pl)
    np.savez_compressed(os.path.join(OUT, "ref_luts.npz"), **luts)
    np.savez_compressed(os.path.join(OUT, "ref_quantize.npz"), **qv)
    np.savez_compressed(os.path.join(OUT, "ref_transform.npz"), **tr)
    for fn in ("ref_luts.npz", "ref_quantize.npz", "ref_transform.npz", "ref_planes.npz", "ref_plane_digests.json"):
        print(fn, os.path.getsize(os.path.join(OUT, fn)), "bytes")


if __name__ == "__main__":
    main()

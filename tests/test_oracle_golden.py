"""The oracle against (a) the committed fixtures that tests/golden/make_golden.py generated from the real
reference LumaQuantizer and (b), where oracle/_ref/libluma_ref.so is present, the reference itself, live.
CPU only; bit-exact everywhere (NaN == NaN)."""
import os

import numpy as np
import pytest

from tests.golden.make_golden import CONFIGS


def same(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def table_for(o, cfg):
    if cfg[0] in (o.PTF_PSI, o.PTF_JND_HDRVDP):
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lumahdrv_amd", "data")
        nm = "psi" if cfg[0] == o.PTF_PSI else "jnd_hdrvdp"
        return np.fromfile(os.path.join(d, "ptf_%s_%d.f32" % (nm, cfg[1])), dtype="<f4")
    return None


@pytest.fixture(scope="module")
def gold(golden_dir):
    return {k: np.load(os.path.join(golden_dir, "ref_%s.npz" % k)) for k in ("luts", "quantize", "transform")}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_lut_matches_reference(oracle_mod, gold, name):
    o = oracle_mod
    cfg = CONFIGS[name]
    m = o.Oracle(*cfg, table=table_for(o, cfg)).mapping
    assert same(m, gold["luts"][name])


@pytest.mark.parametrize("name", list(CONFIGS))
def test_quantize_dequantize_match_reference(oracle_mod, gold, name):
    o = oracle_mod
    cfg = CONFIGS[name]
    qq = o.Oracle(*cfg, table=table_for(o, cfg))
    g = gold["quantize"]
    got0 = np.array([qq.quantize(float(v), 0) for v in g[name + "_in0"]])
    assert np.array_equal(got0.astype(np.uint16), g[name + "_q0"])
    got1 = np.array([qq.quantize(float(v), 1) for v in g[name + "_in1"]])
    assert np.array_equal(got1.astype(np.uint16), g[name + "_q1"])
    codes = np.arange(-2, 2 ** cfg[1] + 2, dtype=np.float32)
    assert same([qq.dequantize(float(c), 0) for c in codes], g[name + "_dq0"])
    ccodes = np.arange(0, 2 ** cfg[3], dtype=np.float32)
    assert same([qq.dequantize(float(c), 1) for c in ccodes], g[name + "_dq1"])


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("sc", [1.0, 20.0, 0.25])
def test_transform_matches_reference(oracle_mod, gold, name, sc):
    o = oracle_mod
    cfg = CONFIGS[name]
    qq = o.Oracle(*cfg, table=table_for(o, cfg))
    t = gold["transform"]
    f = t["input"].copy()
    qq.transform(f, True, sc)
    assert same(f, t["%s_fwd_sc%g" % (name, sc)])
    g = t["%s_inv_in_sc%g" % (name, sc)].copy()
    qq.transform(g, False, sc)
    assert same(g, t["%s_inv_sc%g" % (name, sc)])


def test_live_reference_random_frames(oracle_mod):
    """oracle vs the reference .so on fresh random frames, all colour spaces, both directions"""
    o = oracle_mod
    if not o.have_ref():
        pytest.skip("oracle/_ref/libluma_ref.so not built (needs /root/reference)")
    rng = np.random.default_rng(1234)
    for name, cfg in CONFIGS.items():
        qq = o.Oracle(*cfg, table=table_for(o, cfg))
        r = o.RefQuantizer(*cfg)
        assert same(qq.mapping, r.mapping)
        f = np.exp(rng.uniform(np.log(1e-5), np.log(1e5), size=(3, 32, 64))).astype(np.float32)
        f[:, 0, :4] = [[0, np.nan, np.inf, -3], [1, 1, 1, 2], [2, 1, 1, 1]]
        for sc in (1.0, 3.5):
            a, b = f.copy(), f.copy()
            qq.transform(a, True, sc)
            r.transform(b, True, sc)
            assert same(a, b), (name, sc)
            vals = a[0].ravel()
            assert np.array_equal(np.array([qq.quantize(float(v), 0) for v in vals[:512]]), r.quantize_array(vals[:512], 0))
            assert np.array_equal(np.array([qq.quantize(float(v), 1) for v in a[1].ravel()[:512]]),
                                  r.quantize_array(a[1].ravel()[:512], 1))
            a2, b2 = np.abs(a).copy(), np.abs(a).copy()
            qq.transform(a2, False, sc)
            r.transform(b2, False, sc)
            assert same(a2, b2), (name, sc, "inv")


def test_roundtrip_decode_of_encode(oracle_mod):
    """encode -> decode through the plane layout reproduces the dequantized Lu'v' of every pixel"""
    o = oracle_mod
    qq = o.Oracle(o.PTF_PQ, 11, o.CS_LUV, 8, 1e4, 0.005)
    f = o.synth_frame(64, 32, frame=3)
    src = f.copy()
    for profile in (0, 1, 2, 3):
        qp = o.Oracle(o.PTF_PQ, 8 if profile < 2 else 11, o.CS_LUV, 8, 1e4, 0.005)
        g = src.copy()
        planes, strides, _ = qp.encode(g, 1.0, profile)
        out = qp.decode(planes, strides, 64, 32, 1.0, profile)
        assert out.shape == src.shape and np.all(np.isfinite(out))
        # luminance survives within one quantisation step of the PTF
        Y = 0.212656 * src[0] + 0.715158 * src[1] + 0.072186 * src[2]
        Yd = 0.212656 * out[0] + 0.715158 * out[1] + 0.072186 * out[2]
        assert np.median(np.abs(np.log2(Yd / Y))) < (0.2 if profile < 2 else 0.02)


def test_live_reference_whole_frame_encode(oracle_mod):
    """the oracle's whole-frame encode against the REAL LumaQuantizer driven through the harness's plane loop
    (oracle/ref_harness.cpp: ref_encode_frame), every profile; PSI/HDR-VDP included (tables compiled into the
    reference).  Also reproduces the SURVEY Y-plane digest with the real quantizer."""
    o = oracle_mod
    if not o.have_ref():
        pytest.skip("oracle/_ref/libluma_ref.so not built (needs /root/reference)")
    for name, cfg in CONFIGS.items():
        qq = o.Oracle(*cfg, table=table_for(o, cfg))
        r = o.RefQuantizer(*cfg)
        if not hasattr(r.L, "ref_encode_frame"):
            pytest.skip("stale reference library")
        f = o.synth_frame(96, 40, frame=2)
        f[:, 0, :3] = [[np.nan, 0, -1], [1, 0, 5], [1, 0, 2]]
        for profile in (0, 1, 2, 3):
            a, _, avga = qq.encode(f.copy(), 1.0, profile)
            b, _, avgb = r.encode(f.copy(), 1.0, profile)
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (name, profile)
            assert (np.isnan(avga) and np.isnan(avgb)) or avga == avgb
    r = o.RefQuantizer(o.PTF_PQ, 11, o.CS_LUV, 8, 1e4, 0.005)
    planes, _, _ = r.encode(o.test_frame(1280, 720), 1.0, 2)
    assert o.survey_digest(o.packed_rows(planes[0], 2560)) == "e0ff09731298e8f6"

"""The red / blue tables of the YCbCr decode kernels (k_decode<CS_YCBCR, ., ., ., ., YT, RB>, luma_kernels.hpp; include/lumahip.h
lumahip_rb_table_info) against the CPU oracle -- needs an MI355X.  Bar: decoded floats bit-equal (0 ulp) to the oracle whichever
path a wave takes (tables where its codes are local, six powf per pixel elsewhere), for every lumahip_tune("ycbcr_rb_tables") mode;
and the choice of kernel per launch is a function of the stream's data alone (lumahip_internal.hpp LagPolicy)."""
import numpy as np
import pytest

from tests.test_gpu_half_table import _half_policy_model

pytestmark = pytest.mark.gpu

CFG = (1, 10, 2, 10, 1000.0, 0.01)      # the HDR10 recipe: PQ 10-bit, YCbCr, 10-bit chroma, 1000 / 0.01 cd/m2


def _planes_from_codes(L, y, cb, cr, profile):
    """Y / Cb / Cr code arrays -> planes laid out as vpx_img_alloc does (16-bit little-endian samples for profiles 2, 3)"""
    h, w = y.shape
    _, hs, st, bps = L.plane_geometry(w, h, profile)
    out = []
    for p, c in enumerate((y, cb, cr)):
        buf = np.zeros((hs[p], st[p]), dtype=np.uint8)
        if bps == 2:
            buf[:, :2 * c.shape[1]] = c.astype("<u2").view(np.uint8).reshape(c.shape[0], 2 * c.shape[1])
        else:
            buf[:, :c.shape[1]] = c.astype(np.uint8)
        out.append(buf)
    return out, st


def _picture_codes(rng, h, w, sub, noise=6):
    """smooth code fields with a little noise: what a picture's planes look like (neighbouring pixels, neighbouring codes)"""
    yy, xx = np.mgrid[0:h, 0:w]
    y = 500 + 380 * np.sin(xx / 97.0) * np.cos(yy / 61.0) + rng.normal(0, noise, (h, w))
    ch, cw = (h // 2, w // 2) if sub else (h, w)
    cy, cx = np.mgrid[0:ch, 0:cw]
    cb = 512 + 160 * np.sin(cx / 53.0 + 1.0) + rng.normal(0, noise / 3, (ch, cw))
    cr = 512 + 160 * np.cos(cy / 47.0) + rng.normal(0, noise / 3, (ch, cw))
    return [np.clip(np.rint(a), 0, 1023).astype(np.uint16) for a in (y, cb, cr)]


def _random_codes(rng, h, w, sub, top=1024):
    ch, cw = (h // 2, w // 2) if sub else (h, w)
    return [rng.integers(0, top, s, dtype=np.uint16) for s in ((h, w), (ch, cw), (ch, cw))]


@pytest.mark.parametrize("sc", [20.0, 1.0, 3e30])
@pytest.mark.parametrize("profile", [2, 3])
def test_decoded_floats_equal_the_oracle_whichever_path_runs(oracle_mod, profile, sc):
    """picture-like codes (the waves take the tables), unrelated codes (they do not), codes beyond maxVal / maxC (garbage from a
    lossy upstream decoder: the complete functions), a width that is not a multiple of 4 (two pixels per thread); modes 0 / 1 / 2;
    preScaling 20 (short division), 1 (none) and 3e30 (outside the short division's licence: complete functions throughout)"""
    import lumahdrv_amd as L
    o = oracle_mod
    orc = o.Oracle(*CFG)
    rng = np.random.default_rng(5)
    sub = profile == 2
    cases = []
    for w, h in ((512, 128), (258, 66)):
        cases.append(_picture_codes(rng, h, w, sub))
        cases.append(_random_codes(rng, h, w, sub))
        g = _picture_codes(rng, h, w, sub)
        g[0][::7, ::5] = 1024 + rng.integers(0, 3000, g[0][::7, ::5].shape)      # beyond maxVal: clamped by the table read
        g[1][::3, ::11] = 1024 + rng.integers(0, 60000, g[1][::3, ::11].shape)   # beyond maxC: out of the tables
        g[2][1::5, ::9] = 65535
        cases.append(g)
    for mode in (1, 0, 2):
        q = L.LumaQuantizer()
        q.ctx.tune("ycbcr_rb_tables", mode)
        q.setQuantizer(*CFG)
        info = q.ctx.rb_table_info(sc)
        assert info["used"] == (mode != 0) and (mode == 0 or info["bytes"] == 2 * 1024 * 1024 * 4)
        for y, cb, cr in cases:
            planes, st = _planes_from_codes(L, y, cb, cr, profile)
            h, w = y.shape
            got = q.ctx.decode_frame(planes, st, w, h, sc, profile)
            with np.errstate(all="ignore"):
                exp = orc.decode(planes, st, w, h, sc, profile)
            same = (got.view(np.uint32) == exp.view(np.uint32)) | (np.isnan(got) & np.isnan(exp))
            assert bool(same.all()), (mode, profile, sc, (h, w), np.argwhere(~same)[:4].tolist())
        assert (q.ctx.rb_table_info(sc)["table_launches"] > 0) == (mode != 0)


def test_tables_follow_the_quantizer_and_the_prescaling(oracle_mod):
    """the tables belong to (table, maxLum, bit depths, preScaling): two preScalings alternate (two device copies), a third evicts
    the older one, a new quantizer on the same context rebuilds everything -- every frame still equals the oracle"""
    import lumahdrv_amd as L
    o = oracle_mod
    rng = np.random.default_rng(6)
    y, cb, cr = _picture_codes(rng, 64, 256, True)
    q = L.LumaQuantizer()
    for cfg in (CFG, (1, 10, 2, 10, 4000.0, 0.005), (1, 9, 2, 8, 1000.0, 0.01)):
        q.setQuantizer(*cfg)
        orc = o.Oracle(*cfg)
        top, topc = (1 << cfg[1]) - 1, (1 << cfg[3]) - 1
        planes, st = _planes_from_codes(L, np.minimum(y, top), np.minimum(cb, topc), np.minimum(cr, topc), 2)
        for sc in (20.0, 4.0, 20.0, 0.5, 4.0, 1.0):
            got = q.ctx.decode_frame(planes, st, 256, 64, sc, 2)
            exp = orc.decode(planes, st, 256, 64, sc, 2)
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (cfg, sc)
        assert q.ctx.rb_table_info(1.0)["used"]
    # 12-bit luminance and colour: 2 x 64 MiB of tables, still read (what matters is the L1 hit rate of the lines a picture touches)
    cfg12 = (1, 12, 2, 12, 1000.0, 0.01)
    q.setQuantizer(*cfg12)
    info = q.ctx.rb_table_info(20.0)
    assert info["used"] and info["bytes"] == 2 * (1 << 24) * 4
    orc = o.Oracle(*cfg12)
    y12, cb12, cr12 = [np.minimum(a.astype(np.uint32) * 4 + 1, 4095).astype(np.uint16) for a in (y, cb, cr)]
    planes, st = _planes_from_codes(L, y12, cb12, cr12, 2)
    got = q.ctx.decode_frame(planes, st, 256, 64, 20.0, 2)
    assert np.array_equal(got.view(np.uint32), orc.decode(planes, st, 256, 64, 20.0, 2).view(np.uint32))
    assert q.ctx.rb_table_info(20.0)["table_launches"] > 0
    # 14-bit luminance with 12-bit colour would be 2 x 256 MiB: no tables, the plain kernels
    q.setQuantizer(1, 14, 2, 12, 1000.0, 0.01)
    assert not q.ctx.rb_table_info(20.0)["used"]


def test_no_memory_for_the_tables_means_the_plain_kernels_not_an_error(oracle_mod):
    """the tables are a speed-up, not a requirement: when their allocation fails the decode call goes on with six powf
    per pixel -- same floats --, lumahip_rb_table_info reports "not used" instead of failing, the stream does not ask again, and
    the next lumahip_set_quantizer starts afresh"""
    import lumahdrv_amd as L
    o = oracle_mod
    rng = np.random.default_rng(8)
    y, cb, cr = _picture_codes(rng, 64, 256, True)
    planes, st = _planes_from_codes(L, y, cb, cr, 2)
    orc = o.Oracle(*CFG)
    exp = orc.decode(planes, st, 256, 64, 20.0, 2)
    q = L.LumaQuantizer()
    q.setQuantizer(*CFG)
    q.ctx.tune("test_fail_rb_alloc", 1)                     # the next hipMalloc of the tables "fails"
    got = q.ctx.decode_frame(planes, st, 256, 64, 20.0, 2)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    info = q.ctx.rb_table_info(20.0)
    assert not info["used"] and info["table_launches"] == 0
    got = q.ctx.decode_frame(planes, st, 256, 64, 20.0, 2)  # no retry per launch
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)) and q.ctx.rb_table_info(20.0)["table_launches"] == 0
    q.setQuantizer(*CFG)                                    # a new stream asks again
    got = q.ctx.decode_frame(planes, st, 256, 64, 20.0, 2)
    info = q.ctx.rb_table_info(20.0)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)) and info["used"] and info["table_launches"] == 1


def test_streams_of_unrelated_pixels_back_off_to_the_plain_kernels(oracle_mod):
    """mode 1: a launch none of whose waves found its codes local leaves its feedback word clear; the host reads launch j's word
    when it issues eligible launch j + 4 (after j's completion event) and sends 16 launches to the kernels without the test, then
    ONE launch probes, the pause doubling (up to 64) while probes stay clear.  The launch counts must equal the policy's model (the one of the
    half-input table, tests/test_gpu_half_table.py, with "bad" = no wave gathered) after every launch, synchronising after each
    launch and not at all; every frame equals the oracle."""
    import lumahdrv_amd as L
    o = oracle_mod
    orc = o.Oracle(*CFG)
    rng = np.random.default_rng(8)
    w, h, sc = 512, 64, 20.0
    codes = {False: _picture_codes(rng, h, w, True), True: _random_codes(rng, h, w, True)}
    planes, exp = {}, {}
    for k, (y, cb, cr) in codes.items():
        planes[k], st = _planes_from_codes(L, y, cb, cr, 2)
        exp[k] = orc.decode(planes[k], st, w, h, sc, 2)
    kinds = [False] * 3 + [True] * 150 + [False] * 80 + [True] * 2 + [False] * 30      # (150: the pause reaches its cap of 64)
    model = _half_policy_model(kinds, longest=64)
    sizes = [p.size for p in planes[False]]
    for sync_each in (True, False):
        q = L.LumaQuantizer()
        q.setQuantizer(*CFG)
        c = q.ctx
        d_pl = {k: [c.malloc(n) for n in sizes] for k in planes}
        for k in planes:
            for p in range(3):
                c.h2d(d_pl[k][p], planes[k][p])
        d_out = [c.malloc(3 * w * h * 4) for _ in kinds]
        base = c.rb_table_info(sc)
        assert base["used"] and base["table_launches"] == 0
        for e, k in enumerate(kinds):
            c.decode_frames_device(d_pl[k], st, sizes, 1, w, h, 2, sc, d_out[e], 3 * w * h)
            if sync_each:
                c.sync()
            i = c.rb_table_info(sc)
            assert (i["table_launches"], i["backoff_launches"]) == model[e], (sync_each, e)
        c.sync()
        for e, k in enumerate(kinds):
            got = np.empty((3, h, w), dtype=np.float32)
            c.d2h(got, d_out[e])
            assert np.array_equal(got.view(np.uint32), exp[k].view(np.uint32)), (sync_each, e)
        for b in [x for v in d_pl.values() for x in v] + d_out:
            c.free(b)

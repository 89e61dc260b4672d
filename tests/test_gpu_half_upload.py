"""The half upload of the host encode entry points (lumahip_host.hip xfer_h2d_f16: host frames that hold binary16 values cross
PCIe as halves, k_encode<., ., 4, 3, IN16> widens them) against the CPU oracle -- needs an MI355X.  Bar: the planes of every
entry point are the oracle's bit for bit whether a frame goes up as halves, as floats, or changes its mind half way."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFGS = {"pq11_luv": (1, 11, 0, 8, 1e4, 0.005, 1.0), "pq10_ycbcr": (1, 10, 2, 10, 1000.0, 0.01, 20.0), "pq12_rgb": (1, 12, 1, 8, 1e4, 0.005, 1.0)}


def _pair(L, o, name):
    ptf, bits, cs, bitsC, mx, mn, sc = CFGS[name]
    c = L.Context(0)
    c.set_quantizer(ptf, bits, cs, bitsC, mx, mn, L.build_lut(ptf, bits, mx, mn))
    return c, o.Oracle(ptf, bits, cs, bitsC, mx, mn), sc


def _halves(rng, w, h, lo=1e-3, hi=3e4):
    f = np.exp(rng.uniform(np.log(lo), np.log(hi), size=(3, h, w))).astype(np.float32)
    with np.errstate(over="ignore"):
        return f.astype(np.float16).astype(np.float32)


def _check(planes, exp, what):
    for p in range(3):
        assert np.array_equal(planes[p], exp[p]), (what, p, int(np.sum(planes[p] != exp[p])))


@pytest.mark.parametrize("name", list(CFGS))
@pytest.mark.parametrize("w,h", [(3840, 2160), (1920, 1080), (1280, 720), (260, 34), (258, 34), (4, 2)])
def test_single_frame_calls_upload_halves_and_equal_the_oracle(oracle_mod, name, w, h):
    """lumahip_encode_frame_host: frames above 3 Mpixel go up in row bands, smaller ones in one piece; 258 is not a multiple of 4
    (not eligible: floats); profile 2 and 3; pageable and registered memory; the special values a half can hold."""
    import lumahdrv_amd as L
    o = oracle_mod
    c, orc, sc = _pair(L, o, name)
    c.tune("half_upload", 2)
    rng = np.random.default_rng(w + h)
    f = _halves(rng, w, h)
    if w >= 16:
        f[0, 0, :8] = [0.0, -0.0, np.inf, -np.inf, 65504.0, 2.0 ** -24, -1.5, np.float32(np.float16(0.1))]
    eligible = w % 4 == 0
    for profile in ((2, 3) if w <= 1920 else (2,)):
        before = c.half_upload_info()
        planes, st, mean = c.encode_frame(f, sc, profile)
        with np.errstate(all="ignore"):
            exp, _, avg = orc.encode(f.copy(), sc, profile, threads=8)
        _check(planes, exp, (name, w, h, profile))
        after = c.half_upload_info()
        assert (after["half_frames"] - before["half_frames"] == 1) == eligible and after["float_fallbacks"] == before["float_fallbacks"]
        if np.isfinite(avg):
            assert mean == pytest.approx(avg, rel=5e-3)     # (the oracle adds in fp32, band by band; the kernel statistic is the accurate sum)
    if (w, h) == (1920, 1080):
        c.host_register(f)
        planes, st, _ = c.encode_frame(f, sc, 2)
        c.host_unregister(f)
        _check(planes, exp if profile == 2 else orc.encode(f.copy(), sc, 2, threads=8)[0], "registered")
    c.close()


def test_frames_that_stop_being_halves_half_way(oracle_mod):
    """One value that is not a half, placed in the first row band, in a later one, in the last row, in each channel: the bands
    before it went up as halves, the rest of the frame goes up as floats; and the same through the one-piece path of small
    frames.  The next frames are then not tried for a while (lumahip_tune half_upload 1)."""
    import lumahdrv_amd as L
    o = oracle_mod
    c, orc, sc = _pair(L, o, "pq11_luv")
    rng = np.random.default_rng(5)
    for (w, h) in ((3840, 2160), (640, 360)):
        base = _halves(rng, w, h)
        e0, _, _ = orc.encode(base.copy(), sc, 2, threads=8)
        for (ch, y, x) in ((0, 3, 17), (1, h // 2 + 1, 5), (2, h - 1, w - 1), (0, h - 300 if h > 400 else h - 2, 0)):
            c.tune("half_upload", 2)
            f = base.copy()
            f[ch, y, x] = np.float32(1.2345678)
            before = c.half_upload_info()
            planes, _, _ = c.encode_frame(f, sc, 2)
            exp, _, _ = orc.encode(f.copy(), sc, 2, threads=8)
            _check(planes, exp, (w, h, ch, y, x))
            after = c.half_upload_info()
            assert after["float_fallbacks"] == before["float_fallbacks"] + 1
            # ... and a clean frame right after it is uploaded as halves again (mode 2: no pause)
            planes, _, _ = c.encode_frame(base, sc, 2)
            _check(planes, e0, "clean after mixed")
    # the default mode pauses: 16 frames as floats after a miss, then one try
    c.tune("half_upload", 1)
    f = base.copy()
    f[1, 7, 9] = np.float32(0.3)
    b0 = c.half_upload_info()
    c.encode_frame(f, sc, 2)
    assert c.half_upload_info()["float_fallbacks"] == b0["float_fallbacks"] + 1 and c.half_upload_info()["pause_left"] == 16
    for k in range(16):
        planes, _, _ = c.encode_frame(base, sc, 2)
        assert c.half_upload_info()["half_frames"] == b0["half_frames"] and c.half_upload_info()["pause_left"] == 15 - k
    _check(planes, e0, "paused")
    planes, _, _ = c.encode_frame(base, sc, 2)
    _check(planes, e0, "after the pause")
    assert c.half_upload_info()["half_frames"] == b0["half_frames"] + 1
    c.close()


@pytest.mark.parametrize("name", ["pq11_luv", "pq10_ycbcr"])
def test_batched_and_streaming_entry_points(oracle_mod, name):
    """lumahip_encode_frames_host (3-slot pipeline) and lumahip_encode_stream_push / _pop with half-exact frames, frames of
    floats and a mix of both in one batch; the caller's frame may be overwritten as soon as push returns."""
    import lumahdrv_amd as L
    o = oracle_mod
    c, orc, sc = _pair(L, o, name)
    c.tune("half_upload", 2)
    w, h = 1920, 1080
    rng = np.random.default_rng(9)
    frames = [_halves(rng, w, h) for _ in range(5)]
    frames[2] = np.exp(rng.uniform(np.log(1e-3), np.log(1e4), size=(3, h, w))).astype(np.float32)      # floats
    frames[4][2, h - 1, w - 1] = np.float32(0.7)                                                         # halves but for the last value
    exps = [orc.encode(f.copy(), sc, 2, threads=8)[0] for f in frames]
    before = c.half_upload_info()
    planes, st, means = c.encode_frames(frames, sc, 2)
    for i in range(5):
        _check(planes[i], exps[i], ("batched", i))
    after = c.half_upload_info()
    assert after["half_frames"] - before["half_frames"] == 3 and after["float_fallbacks"] - before["float_fallbacks"] == 2
    # streaming: push(i + 1) before pop(i); the staging frame is scribbled over right after every push
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    outs = [[np.zeros((hs[p], st[p]), dtype=np.uint8) for p in range(3)] for _ in range(5)]
    scratch = np.empty_like(frames[0])
    sa = (C.c_int * 3)(*st)

    def push(i):
        scratch[...] = frames[i]
        pp = (C.c_void_p * 3)(*[a.ctypes.data for a in outs[i]])
        c._chk(c.L.lumahip_encode_stream_push(c.h, scratch.ctypes.data_as(C.c_void_p), w, h, C.c_float(sc), 2, pp, sa))
        scratch[...] = -1.0

    c.L.lumahip_encode_stream_push.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_float, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    c.L.lumahip_encode_stream_pop.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    m = C.c_float(0)
    push(0)
    for i in range(1, 5):
        push(i)
        c._chk(c.L.lumahip_encode_stream_pop(c.h, C.byref(m)))
        _check(outs[i - 1], exps[i - 1], ("stream", i - 1))
    c._chk(c.L.lumahip_encode_stream_pop(c.h, C.byref(m)))
    _check(outs[4], exps[4], ("stream", 4))
    c.close()


def test_dark_half_frames_get_the_references_mean(oracle_mod):
    """A frame whose mean luminance lies near the reference's `<= 1` threshold: the host entry points recompute the reference's
    sequential sum from the uploaded frame -- which is binary16 here (k_channel0<., IN16>)."""
    import lumahdrv_amd as L
    o = oracle_mod
    c, orc, sc = _pair(L, o, "pq11_luv")
    c.tune("half_upload", 2)
    rng = np.random.default_rng(21)
    for (w, h) in ((1280, 720), (3840, 2160)):
        f = _halves(rng, w, h, lo=0.2, hi=5.0)
        planes, _, mean = c.encode_frame(f, sc, 2)
        exp, _, avg = orc.encode(f.copy(), sc, 2)          # single-threaded: the reference's summation order
        _check(planes, exp, (w, h))
        assert 0.25 <= avg <= 4 and np.float32(mean) == np.float32(avg), (mean, avg)
    assert c.half_upload_info()["half_frames"] == 2
    c.close()

"""The half-input table of the YCbCr encode kernels (k_encode<CS_YCBCR, ., ., 6>, luma_device.hpp half_lookup) against the CPU
oracle -- needs an MI355X.  Bar: Y / Cb / Cr planes bit-exact for every binary16 input, for inputs that are not halves, and
for any mix of the two inside one wave."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _encode_device(c, L, frame, sc, profile):
    """one frame through lumahip_encode_frames_device WITHOUT per-frame statistics -- the launch that takes the half-input kernel"""
    _, h, w = frame.shape
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    sizes = [hs[p] * st[p] for p in range(3)]
    d_src = c.malloc(frame.nbytes)
    d_pl = [c.malloc(s) for s in sizes]
    try:
        c.h2d(d_src, frame)
        for p in range(3):
            c.h2d(d_pl[p], np.zeros(sizes[p], dtype=np.uint8))   # the oracle's planes have zeros in the stride padding
        c.encode_frames_device(d_src, 3 * w * h, 1, w, h, sc, profile, d_pl, st, sizes)
        c.sync()
        out = []
        for p in range(3):
            a = np.empty((hs[p], st[p]), dtype=np.uint8)
            c.d2h(a, d_pl[p])
            out.append(a)
    finally:
        c.free(d_src)
        for p in d_pl:
            c.free(p)
    return out, st


def _all_halves_frame(rng):
    """512 x 384 = 3 x 65536 pixels.  Every one of the 65536 binary16 patterns (negative values, -0, denormals, +-inf, NaNs of
    both signs) appears in every colour channel: third 1 is grey (r = g = b = the half), third 2 pairs every half in R with
    random finite non-negative halves in G and B, third 3 permutes all patterns independently per channel."""
    allh = np.arange(65536, dtype=np.uint16)
    pos = np.arange(0x7C00, dtype=np.uint16)
    r = np.concatenate([allh, allh, rng.permutation(allh)])
    g = np.concatenate([allh, rng.choice(pos, 65536), rng.permutation(allh)])
    b = np.concatenate([allh, rng.choice(pos, 65536), rng.permutation(allh)])
    f = np.stack([x.view(np.float16).astype(np.float32) for x in (r, g, b)])
    return np.ascontiguousarray(f.reshape(3, 384, 512))


@pytest.mark.parametrize("max_lum", [1000.0, 1e4])
@pytest.mark.parametrize("sc", [1.0, 20.0, 0.25])
def test_every_half_through_the_half_input_kernel(oracle_mod, sc, max_lum):
    import lumahdrv_amd as L
    o = oracle_mod
    cfg = (L.PTF_PQ, 10, L.CS_YCBCR, 10, max_lum, 0.01)
    q = L.LumaQuantizer()
    q.setQuantizer(*cfg)
    orc = o.Oracle(*cfg)
    c = q.ctx
    c.tune("half_table", 2)          # always: the NaN patterns of this frame would otherwise look like a float stream (see the back-off test)
    info = c.half_table_info(sc)
    assert info["used"] and info["entries"] == 0x7C01 and 124 * 1024 < info["lds_bytes"] <= 160 * 1024, info
    f = _all_halves_frame(np.random.default_rng(int(sc * 4) + int(max_lum)))
    for profile in (3, 2, 1, 0):     # 4:4:4 first: there every pixel's Cb / Cr is a sample of its own
        got, st = _encode_device(c, L, f, sc, profile)
        with np.errstate(all="ignore"):
            exp, est, _ = orc.encode(f.copy(), sc, profile)
        assert tuple(st) == tuple(est)
        for p in range(3):
            assert np.array_equal(got[p], exp[p]), (sc, max_lum, profile, p, int(np.sum(got[p] != exp[p])))
    # the same launch with the table turned off: the per-pixel kernels, same planes
    c.tune("half_table", 0)
    assert not c.half_table_info(sc)["used"]
    off, _ = _encode_device(c, L, f, sc, 2)
    c.tune("half_table", 2)
    on, _ = _encode_device(c, L, f, sc, 2)
    for p in range(3):
        assert np.array_equal(off[p], on[p])


@pytest.mark.parametrize("w,h", [(1024, 256), (258, 34), (4, 2)])
def test_mixed_half_and_float_inputs(oracle_mod, w, h):
    """Inputs that are halves and inputs that are not, mixed at every granularity the kernel has: whole tiles, whole waves,
    single lanes, a single channel of a single pixel.  A lane whose unit (4 x 2 pixels) holds anything but halves takes the
    general path inside the same launch."""
    import lumahdrv_amd as L
    o = oracle_mod
    cfg = (L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01)
    q = L.LumaQuantizer()
    q.setQuantizer(*cfg)
    q.ctx.tune("half_table", 2)
    orc = o.Oracle(*cfg)
    rng = np.random.default_rng(w * 7 + h)
    full = np.exp(rng.uniform(np.log(1e-6), np.log(7e4), size=(3, h, w))).astype(np.float32)
    with np.errstate(over="ignore"):
        halfx = full.astype(np.float16).astype(np.float32)       # overflow -> inf: still a half
    for pattern in ("rows", "blocks", "pixels", "one_channel", "all_float", "all_half"):
        m = np.zeros((3, h, w), dtype=bool)                      # True = keep the full-precision float
        if pattern == "rows":
            m[:, (np.arange(h) // 2) % 2 == 1, :] = True
        elif pattern == "blocks":
            m[:, :, (np.arange(w) // 64) % 3 == 0] = True
        elif pattern == "pixels":
            m[:] = (rng.random((h, w)) < 0.01)[None]
        elif pattern == "one_channel":
            m[1] = rng.random((h, w)) < 0.002
        elif pattern == "all_float":
            m[:] = True
        f = np.where(m, full, halfx).astype(np.float32)
        if w >= 64:
            f[0, 0, 5] = np.float32("nan")
            f[2, h - 1, w - 3] = -np.float32(0.37)               # negative, not a half
            f[1, 1, 9] = np.float32(1e-42)                       # fp32 denormal
        for sc in (20.0, 1.0):
            assert q.ctx.half_table_info(sc)["used"]
            for profile in (2, 3):
                got, st = _encode_device(q.ctx, L, f, sc, profile)
                with np.errstate(all="ignore"):
                    exp, _, _ = orc.encode(f.copy(), sc, profile)
                for p in range(3):
                    assert np.array_equal(got[p], exp[p]), (pattern, sc, profile, p, int(np.sum(got[p] != exp[p])))


def test_pairs_without_a_table_and_the_table_cache(oracle_mod):
    """preScalings the table cannot be built for run the per-pixel kernels (and still equal the oracle); a context keeps at
    most four device copies and may be handed a fifth pair at any time."""
    import lumahdrv_amd as L
    o = oracle_mod
    cfg = (L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01)
    q = L.LumaQuantizer()
    q.setQuantizer(*cfg)
    q.ctx.tune("half_table", 2)
    orc = o.Oracle(*cfg)
    f = _all_halves_frame(np.random.default_rng(3))[:, :64, :]
    for sc in (-2.0, 0.0):
        assert not q.ctx.half_table_info(sc)["used"]
        got, _ = _encode_device(q.ctx, L, f, sc, 2)
        with np.errstate(all="ignore"):
            exp, _, _ = orc.encode(f.copy(), sc, 2)
        for p in range(3):
            assert np.array_equal(got[p], exp[p]), (sc, p)
    for i, sc in enumerate((1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 1.0, 20.0)):
        got, _ = _encode_device(q.ctx, L, f, sc, 2)
        with np.errstate(all="ignore"):
            exp, _, _ = orc.encode(f.copy(), sc, 2)
        for p in range(3):
            assert np.array_equal(got[p], exp[p]), (sc, p)
        assert q.ctx.half_table_info(sc)["device_copies"] == min(i + 1, 4)
    # a 12-bit PQ table: whichever kernel the LDS budget allows, the planes are the oracle's
    cfg12 = (L.PTF_PQ, 12, L.CS_YCBCR, 12, 1e4, 0.005)
    q.setQuantizer(*cfg12)
    orc12 = o.Oracle(*cfg12)
    got, _ = _encode_device(q.ctx, L, f, 1.0, 3)
    with np.errstate(all="ignore"):
        exp, _, _ = orc12.encode(f.copy(), 1.0, 3)
    for p in range(3):
        assert np.array_equal(got[p], exp[p]), p


def _half_policy_model(kinds, lag=4, longest=1024):
    """lumahip_core.hip lag_policy_next (LagPolicy, lumahip_internal.hpp) restated: kinds[i] = True when eligible launch i holds full-precision floats (a table launch on
    it reports).  Returns, per launch, (table launches so far, back-off launches so far) AFTER it was issued."""
    ON, BACKOFF, PROBE_WAIT = 0, 1, 2
    state, left, length, pending, table, backoff, out = ON, 0, 0, [], 0, 0, []
    for e, is_float in enumerate(kinds):
        while pending and pending[0][0] + lag <= e:
            _, reported, probe = pending.pop(0)
            if state == ON and reported:
                state, length = BACKOFF, 16
                left = length
            elif state == PROBE_WAIT and probe:
                if reported:
                    length = min(2 * max(length, 8), longest)
                    state, left = BACKOFF, length
                else:
                    state, length = ON, 0
        probe = False
        if state == BACKOFF and left > 0:
            left -= 1
            backoff += 1
        elif state == PROBE_WAIT:
            backoff += 1
        else:
            if state == BACKOFF:
                state, probe = PROBE_WAIT, True
            pending.append((e, is_float, probe))
            table += 1
        out.append((table, backoff))
    return out


def test_float_streams_back_off_to_the_per_pixel_kernels(oracle_mod):
    """lumahip_tune half_table 1 (the default): every table launch reports into its own word whether its pixels were mostly
    full-precision floats; the host reads launch j's word when it issues eligible launch j + 4, after j's completion event.  A
    report sends 16 launches to the per-pixel kernels, then ONE launch probes the table (pause doubling while the probes report
    again, forgotten after a clean probe).  Which kernel a launch takes is therefore a function of the stream's data alone: the
    launch counts must equal the model's after EVERY launch, with a host synchronisation after each launch and with none at
    all.  Every launch equals the oracle whichever kernel ran."""
    import lumahdrv_amd as L
    o = oracle_mod
    cfg = (L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01)
    orc = o.Oracle(*cfg)
    sc, w, h = 20.0, 512, 64
    rng = np.random.default_rng(11)
    fl = np.exp(rng.uniform(np.log(1e-3), np.log(1e4), size=(3, h, w))).astype(np.float32)
    hf = fl.astype(np.float16).astype(np.float32)
    exp = {True: orc.encode(fl.copy(), sc, 2)[0], False: orc.encode(hf.copy(), sc, 2)[0]}
    # halves, a burst of floats (report + 16 per pixel + probe that reports again + 32 per pixel), halves again (clean probe,
    # back on the table), one float launch in the middle of halves, a tail of halves
    kinds = [False] * 2 + [True] * 30 + [False] * 60 + [True] + [False] * 40
    model = _half_policy_model(kinds)
    assert model[1] == (2, 0) and model[5] == (6, 0) and model[6] == (6, 1) and model[21] == (6, 16) and model[22] == (7, 16)
    assert model[26] == (7, 20) and model[-1][0] > 40           # (what the docstring says, spelled out once)
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    sizes = [hs[p] * st[p] for p in range(3)]
    for sync_each in (True, False):
        q = L.LumaQuantizer()
        q.setQuantizer(*cfg)
        c = q.ctx
        assert c.half_table_info(sc)["used"]
        d_src = {k: c.malloc(fl.nbytes) for k in (True, False)}
        c.h2d(d_src[True], fl)
        c.h2d(d_src[False], hf)
        d_pl = [[c.malloc(n) for n in sizes] for _ in kinds]            # every launch has its own planes
        for pl in d_pl:
            for p in range(3):
                c.h2d(pl[p], np.zeros(sizes[p], dtype=np.uint8))
        for e, k in enumerate(kinds):
            c.encode_frames_device(d_src[k], 3 * w * h, 1, w, h, sc, 2, d_pl[e], st, sizes)
            if sync_each:
                c.sync()
            i = c.half_table_info(sc)
            assert (i["table_launches"], i["backoff_launches"]) == model[e], (sync_each, e)
        c.sync()
        for e, k in enumerate(kinds):
            for p in range(3):
                got = np.empty((hs[p], st[p]), dtype=np.uint8)
                c.d2h(got, d_pl[e][p])
                assert np.array_equal(got, exp[k][p]), (sync_each, e, p)
        n_table, n_back = model[-1]
        c.tune("half_table", 2)                                         # always: no reports, no pauses
        c.encode_frames_device(d_src[True], 3 * w * h, 1, w, h, sc, 2, d_pl[0], st, sizes)
        c.sync()
        i = c.half_table_info(sc)
        assert (i["table_launches"], i["backoff_launches"]) == (n_table + 1, n_back)
        for b in list(d_src.values()) + [x for pl in d_pl for x in pl]:
            c.free(b)

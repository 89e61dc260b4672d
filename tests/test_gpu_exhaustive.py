"""Exhaustive statements that only a GPU makes affordable.

1. The luminance search the kernels ship with -- threshold records keyed by float bits (lut_index.hpp / quantize_thresh) and,
   for evenly spaced tables (PTF_LINEAR from 12 bits), keyed by value (LinIndex / quantize_linkey) -- returns the
   same code as the reference's bisection + nearest-of-two (src/luma_quantizer.cpp:222-235) for EVERY fp32 bit
   pattern (all 2^32: zeros, denormals, negatives, +-inf, every NaN payload), for each shipped transfer function,
   through three different instantiations:
     a. the array kernel k_quantize_array (public LumaQuantizer::quantize over arrays; N = 1, explicit NaN test);
     b. the ENCODE KERNEL ITSELF, k_encode<CS_RGB, 4:4:4, VW=4, records> (profile 3; RGB hands raw floats to the
        search for all three planes; N = 4 and N = 8 call shapes, explicit NaN test);
     c. quantize_lut<records, 4, NONNEG=true>, the instantiation the Lu'v' encode kernels call (they promise the
        search "v >= 0 or NaN"), over 0 .. 0x7fffffff = every non-negative float, +inf and every sign-clear NaN, plus
        every sign-set NaN 0xff800001 .. 0xffffffff.
   The other side of each comparison is the literal bisection kernel on the GPU (lumahip_tune "force_literal"), itself
   pinned against the oracle / the reference fixtures in test_gpu_parity.py and spot-checked here against the oracle.
2. The device powf (pow_glibc.hpp) equals the host libm powf for every non-negative float, for the four PQ exponents.
3. The YCbCr decode kernel (8 straight-line powf per pixel with range arguments stated in luma_device.hpp) equals the
   oracle for EVERY (Y, Cb, Cr) code triple of the HDR10 recipe: 1024^3 pixels through 4:4:4 16-bit planes.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (1, 13): 136 KiB of records, one workgroup per CU; (4, 12) / (4, 14): PTF_LINEAR, value-keyed records (search mode 7: 32 / 128 KiB)
TABLES = [(1, 11), (1, 10), (2, 12), (4, 12), (0, 11), (3, 12), (1, 8), (1, 12), (1, 13), (4, 14)]


def _pair(L, ptf, bits, cs):
    """(records context, literal context) for the same table"""
    lut = L.build_lut(ptf, bits, 1e4, 0.005)
    fast = L.Context(0)
    fast.set_quantizer(ptf, bits, cs, 8, 1e4, 0.005, lut)
    assert fast.quantizer_info()["mode"] == (7 if ptf == 4 and bits >= 12 else 3)
    lit = L.Context(0)
    lit.tune("force_literal", 1)
    lit.set_quantizer(ptf, bits, cs, 8, 1e4, 0.005, lut)
    assert lit.quantizer_info()["mode"] in (0, 2)
    return fast, lit


def _oracle_for(o, ptf, bits, cs):
    tab = None
    if ptf in (0, 3):
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lumahdrv_amd", "data")
        tab = np.fromfile(os.path.join(d, "ptf_%s_%d.f32" % ("psi" if ptf == 0 else "jnd_hdrvdp", bits)), dtype="<f4")
    return o.Oracle(ptf, bits, cs, 8, 1e4, 0.005, table=tab)


@pytest.mark.parametrize("ptf,bits", TABLES)
def test_record_search_equals_bisection_for_every_float(oracle_mod, ptf, bits):
    """(a): k_quantize_array, all 2^32 bit patterns"""
    import torch
    import lumahdrv_amd as L
    o = oracle_mod
    dev = torch.device("cuda:0")
    fast, lit = _pair(L, ptf, bits, L.CS_LUV)
    s = torch.cuda.current_stream().cuda_stream
    fast.set_stream(s)
    lit.set_stream(s)
    n = 1 << 27
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    base = torch.arange(n, dtype=torch.int64, device=dev)
    orc = _oracle_for(o, ptf, bits, o.CS_LUV)
    bad = 0
    for chunk in range(32):
        bits32 = (base + chunk * n).to(torch.int32) if chunk < 16 else (base + chunk * n - (1 << 32)).to(torch.int32)
        x = bits32.view(torch.float32)
        fast.quantize_array_device(x.data_ptr(), a.data_ptr(), n, 0)
        lit.quantize_array_device(x.data_ptr(), b.data_ptr(), n, 0)
        bad += int((a != b).sum().item())
        if chunk in (7, 8, 15, 31):   # spot-check the literal kernel against the oracle on a strided sample
            idx = torch.arange(0, n, 65537, device=dev)
            xs = x[idx].cpu().numpy()
            got = b[idx].cpu().numpy()
            exp = np.array([orc.quantize(float(v), 0) for v in xs], dtype=np.float32)
            assert np.array_equal(got, exp)
    assert bad == 0
    fast.set_stream(None)
    lit.set_stream(None)


@pytest.mark.parametrize("ptf,bits", [(1, 11), (2, 12), (1, 12), (1, 8), (4, 12)])
def test_encode_kernel_search_equals_bisection_for_every_float(oracle_mod, ptf, bits):
    """(b): the fused encode kernel, k_encode<CS_RGB, 4:4:4, VW=4, records> (LINEAR-12: value-keyed records, LM = 7).  Frames of 8192 x 4096 pixels whose three
    planes hold consecutive bit patterns (3 x 2^25 per frame, 43 frames cover 2^32 with wrap-around); profile 3
    (16-bit 4:4:4) writes one code per input float.  PQ-11 / LOG-12 are BASELINE configs 1 and 4's tables; PQ-8 goes
    through profile 1 (8-bit samples)."""
    import torch
    import lumahdrv_amd as L
    o = oracle_mod
    dev = torch.device("cuda:0")
    fast, lit = _pair(L, ptf, bits, L.CS_RGB)
    s = torch.cuda.current_stream().cuda_stream
    fast.set_stream(s)
    lit.set_stream(s)
    profile = 3 if bits > 8 else 1
    bps = 2 if profile == 3 else 1
    w, h = 8192, 4096
    npx = w * h
    n3 = 3 * npx
    stride = w * bps
    psz = h * stride
    base = torch.arange(n3, dtype=torch.int64, device=dev)
    pa = [torch.zeros(psz, dtype=torch.uint8, device=dev) for _ in range(3)]
    pb = [torch.zeros(psz, dtype=torch.uint8, device=dev) for _ in range(3)]
    orc = _oracle_for(o, ptf, bits, o.CS_RGB)
    nframes = -(-(1 << 32) // n3)
    bad = 0
    for f in range(nframes):
        bits64 = (base + f * n3) & 0xFFFFFFFF
        bits32 = torch.where(bits64 >= (1 << 31), bits64 - (1 << 32), bits64).to(torch.int32)
        x = bits32.view(torch.float32)
        fast.encode_frames_device(x.data_ptr(), n3, 1, w, h, 1.0, profile, [t.data_ptr() for t in pa], [stride] * 3, [psz] * 3)
        lit.encode_frames_device(x.data_ptr(), n3, 1, w, h, 1.0, profile, [t.data_ptr() for t in pb], [stride] * 3, [psz] * 3)
        for p in range(3):
            bad += int((pa[p] != pb[p]).sum().item())
        if f in (0, 10, 21, 42):      # literal kernel vs oracle on a strided sample of plane 0
            idx = torch.arange(0, npx, 131071, device=dev)
            xs = x[:npx][idx].cpu().numpy()
            codes = pb[0].view(torch.int16)[idx].cpu().numpy().astype(np.int64) & 0xFFFF if bps == 2 else pb[0][idx].cpu().numpy().astype(np.int64)
            exp = np.array([orc.quantize(float(v), 0) for v in xs], dtype=np.int64)
            assert np.array_equal(codes, exp if bps == 2 else exp & 0xFF)
    assert bad == 0
    fast.set_stream(None)
    lit.set_stream(None)


@pytest.mark.parametrize("ptf,bits", [(1, 13), (2, 14), (1, 16)])
def test_global_memory_records_sampled(oracle_mod, ptf, bits):
    """Tables whose records do not fit LDS (here: the LDS limit forced to 0): records in global memory (mode 4) where
    the table qualifies, against the literal bisection on the global-memory table (slow: 13-16 dependent loads per
    value), on 2^22 consecutive bit patterns from every 2^27 boundary plus 2^26 random patterns."""
    import torch
    import lumahdrv_amd as L
    dev = torch.device("cuda:0")
    lut = L.build_lut(ptf, bits, 1e4, 0.005)
    fast = L.Context(0)
    fast.tune("lds_table_max_kb", 0)
    fast.set_quantizer(ptf, bits, L.CS_LUV, 8, 1e4, 0.005, lut)
    if fast.quantizer_info()["mode"] != 4:
        pytest.skip("table does not qualify for records (mode %d)" % fast.quantizer_info()["mode"])
    lit = L.Context(0)
    lit.tune("force_literal", 1)
    lit.set_quantizer(ptf, bits, L.CS_LUV, 8, 1e4, 0.005, lut)
    assert lit.quantizer_info()["mode"] == 2
    s = torch.cuda.current_stream().cuda_stream
    fast.set_stream(s)
    lit.set_stream(s)
    n = 1 << 22
    a = torch.empty(1 << 26, dtype=torch.float32, device=dev)
    b = torch.empty(1 << 26, dtype=torch.float32, device=dev)
    base = torch.arange(n, dtype=torch.int64, device=dev)
    bad = 0
    for chunk in range(32):
        v = (base + chunk * (1 << 27)) & 0xFFFFFFFF
        x = torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32).view(torch.float32)
        fast.quantize_array_device(x.data_ptr(), a.data_ptr(), n, 0)
        lit.quantize_array_device(x.data_ptr(), b.data_ptr(), n, 0)
        bad += int((a[:n] != b[:n]).sum().item())
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.randint(-(1 << 31), (1 << 31) - 1, (1 << 26,), dtype=torch.int64, device=dev, generator=g).to(torch.int32).view(torch.float32)
    fast.quantize_array_device(x.data_ptr(), a.data_ptr(), 1 << 26, 0)
    lit.quantize_array_device(x.data_ptr(), b.data_ptr(), 1 << 26, 0)
    bad += int((a != b).sum().item())
    assert bad == 0
    fast.set_stream(None)
    lit.set_stream(None)


@pytest.mark.parametrize("ptf,bits", TABLES)
def test_luv_kernel_search_variant_for_every_nonnegative_float_and_every_nan(oracle_mod, ptf, bits):
    """(c): quantize_lut<records, 4, NONNEG=true> -- what k_encode<CS_LUV, ...> calls for a row of luminances -- over
    every bit pattern 0 .. 0x7fffffff and every sign-set NaN, against the literal kernel's general instantiation."""
    import torch
    import lumahdrv_amd as L
    dev = torch.device("cuda:0")
    fast, lit = _pair(L, ptf, bits, L.CS_LUV)
    s = torch.cuda.current_stream().cuda_stream
    fast.set_stream(s)
    lit.set_stream(s)
    n = 1 << 28
    a = torch.empty(n, dtype=torch.int16, device=dev)
    b = torch.empty(n, dtype=torch.int16, device=dev)
    bad = 0
    for chunk in range(8):
        fast.quantize_probe_device(a.data_ptr(), chunk * n, n, nonneg=True)
        lit.quantize_probe_device(b.data_ptr(), chunk * n, n, nonneg=False)
        bad += int((a != b).sum().item())
    assert bad == 0
    # sign-set NaNs: 0xff800001 .. 0xffffffff (start at 0xff800004 to keep the 4-value groups aligned, the first three
    # are covered by the general sweep of the other tests) -- all must give maxVal
    m = (1 << 32) - 0xff800004
    fast.quantize_probe_device(a.data_ptr(), 0xff800004, m, nonneg=True)
    assert bool((a[:m] == ((1 << bits) - 1)).all().item())
    # and the general instantiation of the same shape on the negative half, sampled
    for chunk in (8, 11, 15):
        fast.quantize_probe_device(a.data_ptr(), chunk * n, 1 << 24, nonneg=False)
        lit.quantize_probe_device(b.data_ptr(), chunk * n, 1 << 24, nonneg=False)
        bad += int((a[:1 << 24] != b[:1 << 24]).sum().item())
    assert bad == 0
    fast.set_stream(None)
    lit.set_stream(None)


def test_device_powf_matches_host_libm_dense_sweep(oracle_mod):
    """YCbCr forward transform exercised as a powf probe: for grey pixels (r=g=b=v) channel 0 of the GPU transform
    must equal the oracle's (which calls the host libm) for a dense sweep of v -- 2^22 values spanning every
    exponent the PQ path can see, including zero, denormals, huge values, inf and NaN."""
    import lumahdrv_amd as L
    o = oracle_mod
    cfg = (L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01)
    q = L.LumaQuantizer()
    q.setQuantizer(*cfg)
    orc = o.Oracle(*cfg)
    n = 1 << 22
    bits = (np.arange(n, dtype=np.uint64) * ((0x7F800000 + 4096) // n)).astype(np.uint32)
    v = bits.view(np.float32)
    h, w = 1 << 10, 1 << 12
    f = np.stack([v.reshape(h, w)] * 3).copy()
    g = f.copy()
    assert q.transformColorSpace(f, True, 20.0)
    orc.transform(g, True, 20.0)
    same = (f.view(np.uint32) == g.view(np.uint32)) | (np.isnan(f) & np.isnan(g))
    assert same.all()
    # and back (decode direction: 8 more powf per pixel) on in-range channel values
    rng = np.random.default_rng(0)
    c = rng.uniform(0, 1, (3, 256, 512)).astype(np.float32)
    c[0] = rng.uniform(0, 1000, (256, 512)).astype(np.float32)
    d = c.copy()
    assert q.transformColorSpace(c, False, 20.0)
    orc.transform(d, False, 20.0)
    assert ((c.view(np.uint32) == d.view(np.uint32)) | (np.isnan(c) & np.isnan(d))).all()


@pytest.mark.parametrize("regular", [True, False])
def test_device_powf_equals_host_libm_for_every_nonnegative_float(oracle_mod, regular):
    """pow_glibc.hpp on the GPU vs this host's libm powf, bit for bit, for EVERY non-negative fp32 (0, denormals,
    normals, +inf, the NaNs up to 0x7fffffff) and every 257th negative pattern, for each of the four exponents of
    LumaQuantizer::transformPQ (src/luma_quantizer.cpp:485-501) -- 4 x 2.15e9 arguments.  `regular` = the branch-free
    form + fallback that the kernels actually execute."""
    import torch
    import lumahdrv_amd as L
    o = oracle_mod
    dev = torch.device("cuda:0")
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    m, n_ = np.float32(78.8438), np.float32(0.1593)
    ys = [float(n_), float(m), float(np.float32(1.0) / m), float(np.float32(1.0) / n_)]
    n = 1 << 27
    out = torch.empty(n, dtype=torch.float32, device=dev)
    total_bad = 0
    for y in ys:
        for chunk in range(16):                      # 16 x 2^27 = every pattern 0 .. 0x7fffffff
            ctx.powf_probe_device(out.data_ptr(), chunk * n, n, y, regular)
            bad, fb = o.powf_compare(out.cpu().numpy(), chunk * n, y)
            assert bad == 0, ("y=%r first mismatch at bits 0x%08x" % (y, fb))
            total_bad += bad
    # negative half, sampled: a chunk starting at every 2^27 boundary, 2^20 patterns each
    for y in ys:
        for chunk in range(16, 32):
            ctx.powf_probe_device(out.data_ptr(), chunk * n, 1 << 20, y, regular)
            bad, fb = o.powf_compare(out[:1 << 20].cpu().numpy(), chunk * n, y)
            assert bad == 0, ("y=%r first mismatch at bits 0x%08x" % (y, fb))
    assert total_bad == 0


def test_folded_powf_equals_host_libm_for_every_argument_of_its_range(oracle_mod):
    """pow_glibc.hpp powf_folded (round 6: 13 fp64 operations instead of 17 for the two PQ powers whose arguments come from one
    narrow range) on the GPU vs this host's libm powf, bit for bit, for EVERY float of those ranges and beyond them as far as the
    tables reach (the binades split at OFF = 0.69921875): val^(1/m) for val in [2^-21, 1] (PQdec's first power after the reference's
    clamp; every float of [OFF 2^-21, 2 OFF) is swept) and q^m for q in [OFF, 2 OFF) (PQenc's second power lives in [0.8359, 1.0088])."""
    import torch
    import lumahdrv_amd as L
    o = oracle_mod
    dev = torch.device("cuda:0")
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    m = np.float32(78.8438)

    def bits(v):
        return int(np.float32(v).view(np.uint32))
    for y, lo, hi in ((float(np.float32(1.0) / m), bits(0.69921875 * 2.0 ** -21), bits(1.3984375) - 1), (float(m), bits(0.69921875), bits(1.3984375) - 1)):
        n = 1 << 26
        out = torch.empty(n, dtype=torch.float32, device=dev)
        first, swept = lo, 0
        while first <= hi:
            cnt = min(n, hi - first + 1)
            cnt4 = (cnt + 3) // 4 * 4                        # (the compare helper walks whole words; the last word's padding is not compared)
            ctx.powf_probe_device(out.data_ptr(), first, cnt4, y, 2)
            bad, fb = o.powf_compare(out[:cnt].cpu().numpy(), first, y)
            assert bad == 0, ("y=%r first mismatch at bits 0x%08x" % (y, fb))
            first += cnt
            swept += cnt
        assert swept == hi - lo + 1 and swept > (180_000_000 if y < 1 else 8_000_000)
    ctx.set_stream(None)
    ctx.close()


def test_ycbcr_decode_every_code_triple_of_the_hdr10_recipe(oracle_mod):
    """(3): PQ 10-bit / YCbCr 10-bit chroma, max 1000 cd/m2, preScaling 20 (README.md:46-59 of the reference; BASELINE
    configs[2]).  Frame k holds Y code k everywhere, Cb = row index, Cr = column index; profile 3 (4:4:4) so that every
    pixel carries its own triple.  Decoded floats must equal the oracle's bit for bit (0 ulp)."""
    import lumahdrv_amd as L
    o = oracle_mod
    cfg = (o.PTF_PQ, 10, o.CS_YCBCR, 10, 1000.0, 0.01)
    q = L.LumaQuantizer()
    q.setQuantizer(*cfg)
    orc = o.Oracle(*cfg)
    w = h = 1024
    _, hs, st, bps = L.plane_geometry(w, h, 3)
    assert bps == 2 and tuple(hs) == (h, h, h)

    def plane(codes):
        buf = np.zeros((h, st[0]), dtype=np.uint8)
        buf[:, :2 * w] = codes.astype("<u2").view(np.uint8).reshape(h, 2 * w)
        return buf

    cb = plane(np.repeat(np.arange(h, dtype=np.uint16)[:, None], w, axis=1))
    cr = plane(np.repeat(np.arange(w, dtype=np.uint16)[None, :], h, axis=0))
    nthreads = os.cpu_count() or 1
    # the decoder clamps codes above maxVal, so 1024 (one past the last code) is included as well
    for k in range(1025):
        planes = [plane(np.full((h, w), k, dtype=np.uint16)), cb, cr]
        got = q.ctx.decode_frame(planes, st, w, h, 20.0, 3)
        exp = orc.decode(planes, st, w, h, 20.0, 3, threads=nthreads)
        same = (got.view(np.uint32) == exp.view(np.uint32)) | (np.isnan(got) & np.isnan(exp))
        assert bool(same.all()), (k, np.argwhere(~same)[:4].tolist())


@pytest.mark.parametrize("ptf,bits,mx", [(1, 10, 1000.0), (1, 11, 1e4), (1, 12, 1e4), (2, 12, 1e4)])
def test_ycbcr_composite_records_equal_the_reference_arithmetic_for_every_float(ptf, bits, mx):
    """YCbCr encode: the luminance code is read from threshold records of the composite function
    t = 219 y + 16 -> search(PQdec(t / 255)) instead of two powf, a division and the table search per pixel.  Every float
    t in [16, +inf] and every NaN of either sign through the kernels' own lookup (quantize_thresh<4, NONNEG> on the LDS copy)
    against the reference's arithmetic evaluated on the device with the complete powf, IEEE division and the literal bisection
    -- whose equality with the host libm / the oracle the other tests of this file and tests/test_gpu_parity.py establish."""
    import torch
    import lumahdrv_amd as L
    dev = torch.device("cuda:0")
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(ptf, bits, L.CS_YCBCR, 10, mx, 0.005, L.build_lut(ptf, bits, mx, 0.005))
    n = 1 << 26
    a = torch.empty(n, dtype=torch.int16, device=dev)
    b = torch.empty(n, dtype=torch.int16, device=dev)
    bad = 0
    ranges = [(0x41800000, 0x7f800000 + 1 - 0x41800000), (0x7f800000, 1 << 23), (0xff800000, 1 << 23)]   # [16, inf], NaNs, sign-set NaNs (+ -inf)
    for first, count in ranges:
        done = 0
        while done < count:
            m = min(n, (count - done + 3) // 4 * 4)
            fb = (first + done) & 0xFFFFFFFF
            if fb == 0xff800000:
                fb += 4                      # skip -inf itself: t = -inf cannot occur (t >= 16 or NaN)
                m -= 4
            ctx.ycbcr_luma_probe_device(a.data_ptr(), fb, m, direct=False)
            ctx.ycbcr_luma_probe_device(b.data_ptr(), fb, m, direct=True)
            bad += int((a[:m] != b[:m]).sum().item())
            done += n
    assert bad == 0
    assert int(b[:4].max().item()) & 0xFFFF == (1 << bits) - 1        # the last block probed were NaNs: code maxVal
    ctx.set_stream(None)
    ctx.close()

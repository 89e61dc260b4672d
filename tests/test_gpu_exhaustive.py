"""Exhaustive statements that only a GPU makes affordable.

1. The bucketed table search (lut_index.hpp / quantize_lut_bucket) returns the same code as the reference's
   bisection + nearest-of-two for EVERY fp32 bit pattern (all 2^32: zeros, denormals, negatives, +-inf, every
   NaN payload), for each shipped transfer function.  Both sides run on the GPU (LUMAHIP_FORCE_LITERAL selects
   the literal kernel); the literal kernel itself is pinned against the oracle / the reference fixtures in
   test_gpu_parity.py and spot-checked here against the oracle.
2. The device powf (pow_glibc.hpp) equals the host libm powf on a dense sweep for the four PQ exponents
   (the full 2^31 sweep per exponent is tools/verify_powf.cpp, host-side).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ptf,bits", [(1, 11), (1, 10), (2, 12), (4, 12), (0, 11), (3, 12), (1, 8), (1, 12)])
def test_bucketed_search_equals_bisection_for_every_float(oracle_mod, ptf, bits):
    import torch
    import lumahdrv_amd as L
    o = oracle_mod
    dev = torch.device("cuda:0")
    lut = L.build_lut(ptf, bits, 1e4, 0.005)
    os.environ.pop("LUMAHIP_FORCE_LITERAL", None)
    fast = L.Context(0)
    fast.set_quantizer(ptf, bits, L.CS_LUV, 8, 1e4, 0.005, lut)
    assert fast.quantizer_info()["mode"] == 1
    os.environ["LUMAHIP_FORCE_LITERAL"] = "1"
    try:
        lit = L.Context(0)
        lit.set_quantizer(ptf, bits, L.CS_LUV, 8, 1e4, 0.005, lut)
    finally:
        os.environ.pop("LUMAHIP_FORCE_LITERAL", None)
    assert lit.quantizer_info()["mode"] == 0
    s = torch.cuda.current_stream().cuda_stream
    fast.set_stream(s)
    lit.set_stream(s)
    n = 1 << 27
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    base = torch.arange(n, dtype=torch.int64, device=dev)
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
            tab = None
            if ptf in (0, 3):
                d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lumahdrv_amd", "data")
                tab = np.fromfile(os.path.join(d, "ptf_%s_%d.f32" % ("psi" if ptf == 0 else "jnd_hdrvdp", bits)), dtype="<f4")
            orc = o.Oracle(ptf, bits, o.CS_LUV, 8, 1e4, 0.005, table=tab)
            exp = np.array([orc.quantize(float(v), 0) for v in xs], dtype=np.float32)
            assert np.array_equal(got, exp)
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

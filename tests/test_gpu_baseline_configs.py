"""BASELINE.json's five configurations at FULL size against the oracle -- needs an MI355X.

Collected right after tests/test_gpu_parity.py (tests/conftest.py orders the GPU suite: fixture parity first, then this
file, then the sweeps, the facade and tools, and the bench-contract / TSan / placement tests last), so that a driver
run with `-x` has compared every configuration with the reference's results before anything peripheral can stop it.

  C1  test_survey_testframe_digests           ExrInterface::testFrame 1920x1080 (and 1280x720), test_simple_enc parameters;
                                              the digests come from the complete reference encoder
  C2  test_full_size_4k_frame_and_properties  3840x2160 PQ-11 Lu'v' encode + decode, bit-exact, plus size-independent properties
      test_config2_full_500_frame_stream_round_trip   the whole 500-frame stream, encode + decode through all three decode entry points
  C3  test_full_size_other_configs[pq10_ycbcr10]   HDR10 recipe, 4K
  C4  test_full_size_other_configs[log12_luv8]     7680x4320 LOG-12
  C5  test_config5_full_size_stream           the 2000-frame 4K stream, block-sharded over the GPUs the box has
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.golden.make_golden import CONFIGS  # noqa: E402
from tests.test_gpu_parity import L, pair, same_bits  # noqa: E402,F401  (L is the module fixture)
from tests.test_gpu_multi import ROOT, _bench  # noqa: E402


def test_survey_testframe_digests(L, oracle_mod):
    """config C1 (test_simple_enc parameters on ExrInterface::testFrame): Y/U/V digests recorded from the
    complete reference encoder"""
    o = oracle_mod
    q, _ = pair(L, o, CONFIGS["pq11_luv8"])
    for (w, h, d) in [(1280, 720, ("e0ff09731298e8f6", "4c410839cf4228cc", "28868357f4a5e5e5", "db8ff401614db503")),
                      (1920, 1080, ("ccbc4f62ce2708ab", "efe7b8578cae8ef2", "fe374dc25dde5096", "a3e03753f3d1fe44"))]:
        f = o.test_frame(w, h)
        planes, st, mean, tr = q.ctx.encode_frame(f, 1.0, 2, want_transformed=True)
        assert o.survey_digest(o.packed_rows(planes[0], 2 * w)) == d[0]
        assert o.survey_digest(o.packed_rows(planes[1], w)) == d[1]
        assert o.survey_digest(o.packed_rows(planes[2], w)) == d[2]
        assert o.survey_digest(tr) == d[3]   # the in-place Lu'v' floats the reference leaves in the frame
        assert mean > 1.0


def test_full_size_4k_frame_and_properties(L, oracle_mod):
    """BASELINE configs[1] size: one full 3840x2160 frame bit-exact against the (8-thread) oracle, plus
    size-independent properties: synthetic generator parity, launch-geometry independence, and
    encode(decode(planes)) luma idempotence."""
    import torch
    o = oracle_mod
    w, h = 3840, 2160
    q, orc = pair(L, o, CONFIGS["pq11_luv8"])
    dev = torch.device("cuda:0")
    n3 = 3 * w * h
    src = torch.empty(2 * n3, dtype=torch.float32, device=dev)
    q.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    q.ctx.synth_frames_device(src.data_ptr(), n3, 2, w, h, 20250929, 7)
    torch.cuda.synchronize()
    host = src[:n3].cpu().numpy().reshape(3, h, w)
    assert same_bits(host, o.synth_frame(w, h, 20250929, 7))           # device generator == oracle generator
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    sizes = [hs[p] * st[p] for p in range(3)]
    planes = [torch.zeros(2 * sizes[p], dtype=torch.uint8, device=dev) for p in range(3)]
    stats = torch.zeros(6, dtype=torch.float32, device=dev)
    q.ctx.encode_frames_device(src.data_ptr(), n3, 2, w, h, 1.0, 2, [p.data_ptr() for p in planes], st, sizes,
                               stats.data_ptr())
    torch.cuda.synchronize()
    got = [planes[p][:sizes[p]].cpu().numpy().reshape(hs[p], st[p]) for p in range(3)]
    e, _, avg = orc.encode(host.copy(), 1.0, 2, threads=8)
    for p in range(3):
        assert np.array_equal(got[p], e[p]), p
    s = stats.cpu().numpy()
    assert s[0] / (w * h) == pytest.approx(avg, rel=1e-3) and s[1] >= 1e-4 and s[2] <= 1e8
    # decode on device, compare full frame with the oracle (0 ulp)
    out = torch.empty(2 * n3, dtype=torch.float32, device=dev)
    q.ctx.decode_frames_device([p.data_ptr() for p in planes], st, sizes, 2, w, h, 2, 1.0, out.data_ptr(), n3)
    torch.cuda.synchronize()
    dec = out[:n3].cpu().numpy().reshape(3, h, w)
    assert same_bits(dec, orc.decode(e, st, w, h, 1.0, 2, threads=8))
    # idempotence: re-encoding the decoded frame reproduces every luma code whose decoded colour is in gamut
    planes2 = [torch.zeros(2 * sizes[p], dtype=torch.uint8, device=dev) for p in range(3)]
    q.ctx.encode_frames_device(out.data_ptr(), n3, 2, w, h, 1.0, 2, [p.data_ptr() for p in planes2], st, sizes)
    torch.cuda.synchronize()
    y1 = planes[0][:sizes[0]].cpu().numpy().view("<u2").astype(np.int32)
    y2 = planes2[0][:sizes[0]].cpu().numpy().view("<u2").astype(np.int32)
    assert np.mean(np.abs(y1 - y2) <= 1) > 0.999
    q.ctx.set_stream(None)


def test_config2_full_500_frame_stream_round_trip(L, oracle_mod):
    """BASELINE configs[1] as written: the 500-frame 3840x2160 PQ-11 Lu'v' synthetic stream, resident, encoded and decoded in batches
    of 20.  Frames 0, 123, 250 and 499: planes and decoded floats bit-equal to the oracle's.  ALL 500 frames: the decoded stream is the
    same bits through the three decode entry points (packed in one buffer, R / G / B planes in three buffers, packed frames rotating
    over three buffers), and re-encoding the decoded frames reproduces the luminance codes (>= 99.9 % within one code: decoded colours
    outside the gamut clamp, as in the single-frame test above)."""
    import torch
    o = oracle_mod
    w, h, B, F = 3840, 2160, 20, 500
    free = torch.cuda.mem_get_info(0)[0]
    if free < 150e9:
        pytest.skip("needs 150 GB of free HBM")
    q, orc = pair(L, o, CONFIGS["pq11_luv8"])
    c = q.ctx
    dev = torch.device("cuda:0")
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    n1, n3 = w * h, 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    psz = [hs[p] * st[p] for p in range(3)]
    src = torch.empty(F * n3, dtype=torch.float32, device=dev)
    dec = torch.empty(F * n3, dtype=torch.float32, device=dev)
    planes = [torch.zeros(F * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    planes2 = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    alt = torch.empty(B * n3, dtype=torch.float32, device=dev)            # one batch through the other two entry points
    per = -(-B // 3)
    rot = [torch.empty(per * n3, dtype=torch.float32, device=dev) for _ in range(3)]
    close = total = 0
    for b in range(F // B):
        sp, dp = src.data_ptr() + b * B * n3 * 4, dec.data_ptr() + b * B * n3 * 4
        pl = [planes[p].data_ptr() + b * B * psz[p] for p in range(3)]
        c.synth_frames_device(sp, n3, B, w, h, 20250929, b * B)
        c.encode_frames_device(sp, n3, B, w, h, 1.0, 2, pl, st, psz)
        c.decode_frames_device(pl, st, psz, B, w, h, 2, 1.0, dp, n3)
        # the same batch as three colour-plane buffers (channel-major: plane c of frame f at alt + c*B*n1 + f*n1) ...
        c.decode_frames_device_planar(pl, st, psz, B, w, h, 2, 1.0, [alt.data_ptr() + k * B * n1 * 4 for k in range(3)], n1)
        # ... and as packed frames rotating over three buffers
        c.decode_frames_device_rotating(pl, st, psz, B, w, h, 2, 1.0, [r.data_ptr() for r in rot], n3)
        c.encode_frames_device(dp, n3, B, w, h, 1.0, 2, [x.data_ptr() for x in planes2], st, psz)
        torch.cuda.synchronize()
        d = dec[b * B * n3:(b + 1) * B * n3].view(B, 3, n1)
        assert torch.equal(alt.view(3, B, n1).permute(1, 0, 2).view(torch.int32), d.view(torch.int32)), b
        for f in range(B):
            assert torch.equal(rot[f % 3][(f // 3) * n3:(f // 3 + 1) * n3].view(torch.int32), d[f].reshape(-1).view(torch.int32)), (b, f)
        y1 = planes[0][b * B * psz[0]:(b + 1) * B * psz[0]].view(torch.int16).to(torch.int32)
        y2 = planes2[0].view(torch.int16).to(torch.int32)
        close += int(((y1 - y2).abs() <= 1).sum().item())
        total += y1.numel()
    assert close / total > 0.999
    nthreads = min(64, os.cpu_count() or 8)
    for f in (0, 123, 250, 499):
        e, _, _ = orc.encode(o.synth_frame(w, h, 20250929, f), 1.0, 2, threads=nthreads)
        for p in range(3):
            got = planes[p][f * psz[p]:(f + 1) * psz[p]].cpu().numpy().reshape(hs[p], st[p])
            assert np.array_equal(got, e[p]), (f, p)
        assert same_bits(dec[f * n3:(f + 1) * n3].cpu().numpy().reshape(3, h, w), orc.decode(e, st, w, h, 1.0, 2, threads=nthreads)), f
    c.set_stream(None)


@pytest.mark.parametrize("name,w,h,sc", [("log12_luv8", 7680, 4320, 1.0), ("pq10_ycbcr10", 3840, 2160, 20.0)])
def test_full_size_other_configs(L, oracle_mod, name, w, h, sc):
    """BASELINE configs[2] (HDR10 recipe, 4K) and configs[3] (8K LOG-12) at full size: one frame bit-exact both
    ways against the multi-threaded oracle."""
    import torch
    o = oracle_mod
    cfg = CONFIGS[name]
    q, orc = pair(L, o, cfg)
    dev = torch.device("cuda:0")
    n3 = 3 * w * h
    src = torch.empty(n3, dtype=torch.float32, device=dev)
    q.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    q.ctx.synth_frames_device(src.data_ptr(), n3, 1, w, h, 20250929, 3)
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    sizes = [hs[p] * st[p] for p in range(3)]
    planes = [torch.zeros(sizes[p], dtype=torch.uint8, device=dev) for p in range(3)]
    q.ctx.encode_frames_device(src.data_ptr(), n3, 1, w, h, sc, 2, [p.data_ptr() for p in planes], st, sizes)
    out = torch.empty(n3, dtype=torch.float32, device=dev)
    q.ctx.decode_frames_device([p.data_ptr() for p in planes], st, sizes, 1, w, h, 2, sc, out.data_ptr(), n3)
    torch.cuda.synchronize()
    host = src.cpu().numpy().reshape(3, h, w)
    assert same_bits(host, o.synth_frame(w, h, 20250929, 3))
    nthreads = min(64, os.cpu_count() or 8)
    e, _, _ = orc.encode(host.copy(), sc, 2, threads=nthreads)
    for p in range(3):
        assert np.array_equal(planes[p].cpu().numpy().reshape(hs[p], st[p]), e[p]), (name, p)
    assert same_bits(out.cpu().numpy().reshape(3, h, w), orc.decode(e, st, w, h, sc, 2, threads=nthreads)), name
    q.ctx.set_stream(None)


@pytest.mark.parametrize("driver", ["torch", "multi"])
def test_config5_full_size_stream(driver, oracle_mod, tmp_path):
    """BASELINE configs[4] at FULL size on whatever GPUs this box has: the 2000-frame 3840x2160 PQ-11 Lu'v' stream, resident
    (249 GB at N = 1; in consecutive resident blocks where a GPU has less free), block-sharded, through bench.py's one-process-per-GPU driver and through the C ABI's many-GPU layer
    (--driver multi).  The stream digest is independent of the driver and of N and equals the committed one
    (profiles/r02_stream2000_n1.json); four frames are checked against the oracle's planes."""
    import importlib.util
    import torch
    free = min(torch.cuda.mem_get_info(d)[0] for d in range(torch.cuda.device_count()))
    if free < 30e9:
        pytest.skip("needs >= 30 GB of free HBM per GPU (have %.0f GB)" % (free / 1e9))
    # (a GPU that cannot hold its whole shard at once -- 249 GB at N = 1 -- encodes it in consecutive resident blocks: same digest)
    n = torch.cuda.device_count()
    dump = str(tmp_path / "digests.json")
    p = _bench("--gpus", str(n), "--stream-frames", "2000", "--driver", driver, "--min-seconds", "0.2", "--dump-digests", dump)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert r["n_gpus"] == n and r["digests"]["gathered_in_stream_order"] == 2000
    assert r["digests"]["stream_digest"] == "54051a63ee9b1773"
    dig = json.load(open(dump))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    o = oracle_mod
    orc = o.Oracle(o.PTF_PQ, 11, o.CS_LUV, 8, 1e4, 0.005)
    w, h = 3840, 2160
    for f in (0, 777, 1250, 1999):
        planes, st, _ = orc.encode(o.synth_frame(w, h, b.SEED, f), 1.0, 2, threads=os.cpu_count() or 8)
        t = [torch.from_numpy(np.ascontiguousarray(pl).reshape(-1)) for pl in planes]
        psz = [int(x.numel()) for x in t]
        want = int(b.frame_digests(t, psz, 1, torch.device("cpu"))[0].item()) & 0x7FFFFFFFFFFFFFFF
        assert dig[f] == want, "frame %d of the stream differs from the oracle" % f

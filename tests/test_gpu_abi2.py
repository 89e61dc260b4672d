"""ABI version 2 on a real MI355X: planar float frames, unordered sections, the decode traffic probe, the many-GPU layer
(lumahip_multi_* from C++ and through ctypes), the copy threads of the host entry points.  Everything through the C ABI;
the oracle is the checker."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFGS = {"pq11_luv": (1, 11, 0, 8, 1e4, 0.005, 1.0), "pq10_ycbcr": (1, 10, 2, 10, 1000.0, 0.01, 20.0), "log12_xyz": (2, 12, 3, 8, 1e4, 0.005, 1.0)}


def _ctx(L, cfg, torch):
    ptf, bits, cs, bitsC, mx, mn, _ = cfg
    c = L.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_quantizer(ptf, bits, cs, bitsC, mx, mn, L.build_lut(ptf, bits, mx, mn))
    return c


@pytest.mark.parametrize("name", list(CFGS))
@pytest.mark.parametrize("profile", [2, 3])
def test_planar_entry_points_equal_the_packed_ones(oracle_mod, name, profile):
    """float frames as three colour-plane base pointers: channel-major (all R planes, all G planes, all B planes in three
    separate buffers, frame stride w*h), frame-major with padding between the planes, and the packed LumaFrame layout itself --
    identical planes / identical decoded floats to the packed entry points and to the oracle"""
    import torch
    import lumahdrv_amd as L
    o = oracle_mod
    cfg = CFGS[name]
    sc = cfg[6]
    dev = torch.device("cuda:0")
    ctx = _ctx(L, cfg, torch)
    w, h, B = 200, 66, 5                       # w % 4 == 0, ragged against the 256-pixel tiles
    n1, n3 = w * h, 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    src = torch.empty(B * n3, dtype=torch.float32, device=dev)
    ctx.synth_frames_device(src.data_ptr(), n3, B, w, h, 11, 3)
    planes = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    ctx.encode_frames_device(src.data_ptr(), n3, B, w, h, sc, profile, [p.data_ptr() for p in planes], st, psz)
    out = torch.empty(B * n3, dtype=torch.float32, device=dev)
    ctx.decode_frames_device([p.data_ptr() for p in planes], st, psz, B, w, h, profile, sc, out.data_ptr(), n3)
    torch.cuda.synchronize()
    s4 = src.view(B, 3, n1)
    # (a) channel-major: three separate buffers
    chan = [s4[:, k, :].contiguous() for k in range(3)]
    pl_a = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    ctx.encode_frames_device_planar([c.data_ptr() for c in chan], n1, B, w, h, sc, profile, [p.data_ptr() for p in pl_a], st, psz)
    out_a = [torch.full((B, n1), -7.0, dtype=torch.float32, device=dev) for _ in range(3)]
    ctx.decode_frames_device_planar([p.data_ptr() for p in planes], st, psz, B, w, h, profile, sc, [c.data_ptr() for c in out_a], n1)
    # (b) frame-major with 64 floats of padding behind every plane
    pad = n1 + 64
    fm = torch.zeros(B, 3, pad, dtype=torch.float32, device=dev)
    fm[:, :, :n1] = s4
    pl_b = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    ctx.encode_frames_device_planar([fm.data_ptr() + k * pad * 4 for k in range(3)], 3 * pad, B, w, h, sc, profile,
                                    [p.data_ptr() for p in pl_b], st, psz)
    # (c) the packed layout through the planar call
    pl_c = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    ctx.encode_frames_device_planar([src.data_ptr() + k * n1 * 4 for k in range(3)], n3, B, w, h, sc, profile,
                                    [p.data_ptr() for p in pl_c], st, psz)
    torch.cuda.synchronize()
    for other in (pl_a, pl_b, pl_c):
        assert all(torch.equal(a, b) for a, b in zip(planes, other))
    o4 = out.view(B, 3, n1)
    for k in range(3):
        assert torch.equal(out_a[k].view(torch.int32), o4[:, k, :].contiguous().view(torch.int32))
    # against the oracle, frame 2
    orc = o.Oracle(cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5])
    e, _, _ = orc.encode(o.synth_frame(w, h, 11, 3 + 2), sc, profile)
    got = [pl_a[p][2 * psz[p]:3 * psz[p]].cpu().numpy().reshape(hs[p], st[p]) for p in range(3)]
    assert all(np.array_equal(a, b) for a, b in zip(got, e))
    # overlapping planes are refused: G plane starting inside the R planes of a channel-major batch
    with pytest.raises(L.LumaHipError):
        ctx.encode_frames_device_planar([src.data_ptr(), src.data_ptr() + n1 * 4, src.data_ptr() + 2 * n1 * 4], n1, B, w, h, sc,
                                        profile, [p.data_ptr() for p in pl_c], st, psz)
    with pytest.raises(L.LumaHipError):
        ctx.decode_frames_device_planar([p.data_ptr() for p in planes], st, psz, B, w, h, profile, sc,
                                        [out.data_ptr(), out.data_ptr() + (n1 - 4) * 4, out.data_ptr() + 2 * n1 * 4], n3)
    ctx.set_stream(None)
    ctx.close()


@pytest.mark.parametrize("lanes", [1, 2, 3, 4])
def test_unordered_section_gives_the_ordered_results(oracle_mod, lanes):
    """independent batches inside lumahip_begin_unordered / lumahip_end_unordered: same bytes as one stream; work queued on the
    context's stream after `end` sees every batch; misuse is refused"""
    import torch
    import lumahdrv_amd as L
    dev = torch.device("cuda:0")
    cfg = CFGS["pq11_luv"]
    ctx = _ctx(L, cfg, torch)
    w, h, B, NB, profile = 1280, 720, 6, 7, 2
    n3 = 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    src = torch.empty(NB * B * n3, dtype=torch.float32, device=dev)
    ctx.synth_frames_device(src.data_ptr(), n3, NB * B, w, h, 5, 0)

    def run(section):
        planes = [torch.zeros(NB * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
        out = torch.zeros(NB * B * n3, dtype=torch.float32, device=dev)
        if section:
            ctx.begin_unordered(lanes)
        for b in range(NB):
            ctx.encode_frames_device(src.data_ptr() + b * B * n3 * 4, n3, B, w, h, 1.0, profile,
                                     [planes[p].data_ptr() + b * B * psz[p] for p in range(3)], st, psz)
        if section:
            ctx.end_unordered()
            ctx.begin_unordered(lanes)       # decode reads what the encode section wrote: a second section, after the first
        for b in range(NB):
            ctx.decode_frames_device([planes[p].data_ptr() + b * B * psz[p] for p in range(3)], st, psz, B, w, h, profile, 1.0,
                                     out.data_ptr() + b * B * n3 * 4, n3)
        if section:
            ctx.end_unordered()
        chk = out.sum(dtype=torch.float64)   # on torch's current stream = the context's stream: ordered after `end`
        torch.cuda.synchronize()
        return planes, out, float(chk)

    p0, o0, c0 = run(False)
    p1, o1, c1 = run(True)
    assert all(torch.equal(a, b) for a, b in zip(p0, p1)) and torch.equal(o0.view(torch.int32), o1.view(torch.int32)) and c0 == c1
    # misuse
    with pytest.raises(L.LumaHipError):
        ctx.end_unordered()
    ctx.begin_unordered(lanes)
    with pytest.raises(L.LumaHipError):
        ctx.begin_unordered(lanes)
    with pytest.raises(L.LumaHipError):
        ctx.set_stream(None)
    ctx.sync()                               # legal inside a section
    ctx.end_unordered()
    with pytest.raises(L.LumaHipError):
        ctx.begin_unordered(9)
    ctx.set_stream(None)
    ctx.close()


def test_decode_traffic_probe_and_tuning_keys():
    import torch
    import lumahdrv_amd as L
    dev = torch.device("cuda:0")
    ctx = _ctx(L, CFGS["pq11_luv"], torch)
    w, h, B = 1920, 1080, 4
    n1, n3 = w * h, 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    psz = [hs[p] * st[p] for p in range(3)]
    planes = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    out = torch.empty(B * n3, dtype=torch.float32, device=dev)
    ms = ctx.probe_decode_traffic([p.data_ptr() for p in planes], st, psz, B, w, h, [out.data_ptr() + k * n1 * 4 for k in range(3)], n3, iters=3)
    assert ms > 0.0
    assert ctx.device() == 0
    for key, val in (("block", 512), ("block", 0), ("blocks_per_cu", 4), ("blocks_per_cu", 0), ("grid_enc", 512), ("grid_enc", 0),
                     ("lane_grid_dec", 640), ("lane_grid_dec", 0), ("copy_threads", 2), ("lanes", 2), ("lds_table_max_kb", -1)):
        ctx.tune(key, val)
    for key, val in (("block", 100), ("nonsense", 1), ("copy_threads", 99), ("lanes", 17)):
        with pytest.raises(L.LumaHipError):
            ctx.tune(key, val)
    ctx.set_stream(None)
    ctx.close()


def test_search_index_is_lazy_and_cached():
    """a context that only decodes never builds the encode-side search index; quantizer_info (or the first encode) does, and a
    second context with the same table gets the same index (from the process-wide cache).  No timing comparison: the hosts of the
    GPU boxes are shared, and a correctness suite must not depend on wall-clock ordering."""
    import lumahdrv_amd as L
    lut = L.build_lut(L.PTF_PQ, 13)
    c = L.Context(0)
    c.set_quantizer(L.PTF_PQ, 13, L.CS_LUV, 8, 1e4, 0.005, lut)
    info = c.quantizer_info()
    c2 = L.Context(0)
    c2.set_quantizer(L.PTF_PQ, 13, L.CS_LUV, 8, 1e4, 0.005, lut)
    info2 = c2.quantizer_info()
    assert info == info2 and info["mode"] in (3, 4)
    c.close()
    c2.close()


@pytest.mark.parametrize("threads", [0, 1, 4, 7])
def test_copy_threads_do_not_change_results(oracle_mod, threads):
    """pageable host frames staged by 0 / 1 / 4 / 7 copy threads (odd split sizes, strides with padding)"""
    import lumahdrv_amd as L
    o = oracle_mod
    c = L.Context(0)
    c.tune("copy_threads", threads)
    c.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11))
    orc = o.Oracle(o.PTF_PQ, 11, o.CS_LUV, 8, 1e4, 0.005)
    w, h = 2562, 1442                          # 44 MB of floats: several 16 MiB staging chunks, odd row sizes
    f = o.synth_frame(w, h, 3, 1)
    for profile, align in ((2, 32), (3, 96)):
        planes, st, _ = c.encode_frame(f, 1.0, profile, align=align)
        e, st2, _ = orc.encode(f.copy(), 1.0, profile, align=align)
        assert tuple(st2) == tuple(st) and all(np.array_equal(a, b) for a, b in zip(planes, e))
        dec = c.decode_frame(planes, st, w, h, 1.0, profile)
        assert np.array_equal(dec.view(np.uint32), orc.decode(e, st, w, h, 1.0, profile).view(np.uint32))
    c.close()


@pytest.mark.parametrize("bands,taper", [(1, 100), (4, 100), (4, 70), (8, 40), (3, 10), (5, 55)])
def test_row_bands_do_not_change_results(oracle_mod, bands, taper):
    """The single-frame _host entry points cut large frames into tapering row bands (upload | kernel | download overlap):
    every split -- one piece, uniform, tapered, more bands than fit, heights that are not multiples of the 16-row band
    unit -- must give the oracle's bytes, encode and decode, Lu'v' and YCbCr, with the mean luminance of the whole frame."""
    import lumahdrv_amd as L
    o = oracle_mod
    for cfg, sc in (((L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005), 1.0), ((L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01), 20.0)):
        c = L.Context(0)
        c.tune("host_bands", bands)
        c.tune("band_taper", taper)
        c.set_quantizer(*cfg, L.build_lut(cfg[0], cfg[1], cfg[4], cfg[5]))
        orc = o.Oracle(*cfg)
        for (w, h) in ((2562, 1442), (6400, 514), (1538, 2102)):     # >= 3 * 2^20 pixels: banded; 514 rows: fewer bands than asked
            f = o.synth_frame(w, h, 11, bands)
            planes, st, mean = c.encode_frame(f, sc, 2)
            e, st2, emean = orc.encode(f.copy(), sc, 2)
            assert tuple(st2) == tuple(st) and all(np.array_equal(a, b) for a, b in zip(planes, e)), (cfg, w, h)
            assert abs(mean - emean) <= 0.02 * abs(emean) + 1e-6       # (the oracle sums sequentially in fp32, as the reference does)
            dec = c.decode_frame(planes, st, w, h, sc, 2)
            assert np.array_equal(dec.view(np.uint32), orc.decode(e, st, w, h, sc, 2).view(np.uint32)), (cfg, w, h)
        c.close()
    with pytest.raises(L.LumaHipError):
        c2 = L.Context(0)
        try:
            c2.tune("band_taper", 5)
        finally:
            c2.close()


def test_deferred_downloads_do_not_change_results(oracle_mod):
    """pageable planes / frames come back through a ring of pinned chunks that is emptied lazily (when the ring comes round,
    or when the call drains it): batched and single-frame entry points, more frames than pipeline slots, frames larger than
    the ring of download chunks, both directions."""
    import lumahdrv_amd as L
    o = oracle_mod
    c = L.Context(0)
    c.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11))
    orc = o.Oracle(o.PTF_PQ, 11, o.CS_LUV, 8, 1e4, 0.005)
    for (w, h, n, profile) in ((2562, 1442, 7, 2), (1280, 720, 11, 3), (3840, 2160, 4, 2)):
        frames = [o.synth_frame(w, h, 5, i) for i in range(n)]
        got, gst, _ = c.encode_frames(frames, 1.0, profile)
        exp = [orc.encode(f.copy(), 1.0, profile, threads=8) for f in frames]
        st = exp[0][1]
        assert tuple(gst) == tuple(st)
        for i in range(n):
            assert all(np.array_equal(a, b) for a, b in zip(got[i], exp[i][0])), (w, h, i)
        dec = c.decode_frames([e[0] for e in exp], st, w, h, 1.0, profile)
        for i in range(n):
            assert np.array_equal(dec[i].view(np.uint32), orc.decode(exp[i][0], st, w, h, 1.0, profile, threads=8).view(np.uint32)), (w, h, i)
        one, st1, _ = c.encode_frame(frames[-1], 1.0, profile)
        assert all(np.array_equal(a, b) for a, b in zip(one, exp[-1][0]))
        assert np.array_equal(c.decode_frame(one, st1, w, h, 1.0, profile).view(np.uint32), dec[-1].view(np.uint32))
    c.close()


def _build_cpp(tmp, name):
    exe = os.path.join(tmp, name)
    lib = os.path.join(ROOT, "lumahdrv_amd", "lib")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", name + ".cpp"),
                    "-o", exe, "-L" + lib, "-lluma_hip", "-llumahip", "-Wl,-rpath," + lib], check=True)
    return exe


def test_many_gpu_layer_from_cpp(tmp_path):
    """tests/cpp/multi_batch.cpp: a 40-frame stream through lumahip_multi_* over all visible devices and over 2 / 3 logical
    shards on device 0 (host batch, decode, device-resident), frame by frame equal to ONE context looping over the frames as
    the reference's lumaenc does; the table reaches the devices through RCCL; LumaBatchEncoder -> stream -> LumaDecoder"""
    exe = _build_cpp(str(tmp_path), "multi_batch")
    r = subprocess.run([exe, "320", "180", "40", str(tmp_path / "batch.lhs")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "OK all" in r.stdout and "OK 3 shards on device 0" in r.stdout and "OK LumaBatchEncoder" in r.stdout
    # ragged: fewer frames than shards
    r = subprocess.run([exe, "64", "32", "2", str(tmp_path / "few.lhs")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK all" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("w,h,frames,cs,bits,profile", [(1920, 1080, 9, 0, 11, 2), (3840, 2160, 5, 0, 11, 2), (642, 362, 7, 2, 10, 3),
                                                     (7680, 4320, 3, 0, 12, 2)])
def test_pipelined_encoder_writes_the_same_stream(tmp_path, w, h, frames, cs, bits, profile):
    """tests/cpp/pipelined_encoder.cpp: LumaEncoder::setPipelined(true) (lumahip_encode_stream_push / _pop: frame i+1 goes up
    before frame i is completed) writes byte for byte the stream of the synchronous mode, delivers every frame exactly one
    encode() late, lets the caller overwrite its frame at once, and the C ABI refuses what it documents as refused; LumaDecoder::
    setPipelined(true) (lumahip_decode_stream_push / _pop) returns the synchronous decoder's frames bit for bit, in order."""
    exe = _build_cpp(str(tmp_path), "pipelined_encoder")
    r = subprocess.run([exe, str(tmp_path), str(w), str(h), str(frames), str(cs), str(bits), str(profile)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "OK streams identical: %d frames" % frames in r.stdout and "OK stream rules" in r.stdout
    assert "OK pipelined decode: %d frames identical" % frames in r.stdout


def test_many_shard_host_path_randomized_stress(tmp_path):
    """tests/cpp/multi_stress.cpp: 200 iterations of random frame sizes / counts / profiles / strides / pinned-or-pageable buffers
    through 8 logical shards (8 host threads, 8 contexts, 8 copy-thread pools) on ONE GPU, the quantizer replaced every 16
    iterations: planes and decoded floats byte-identical to a single context."""
    exe = _build_cpp(str(tmp_path), "multi_stress")
    r = subprocess.run([exe, "200", "8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "OK multi_stress: 200 iterations, 8 shards" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_many_gpu_layer_through_ctypes_against_the_oracle(oracle_mod):
    import torch
    import lumahdrv_amd as L
    from lumahdrv_amd import capi
    o = oracle_mod
    ndev = torch.cuda.device_count()
    devices = list(range(ndev)) if ndev > 1 else [0, 0, 0]
    m = capi.Multi(devices)
    assert m.shards == len(devices)
    cfg = (L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01)
    m.set_quantizer(*cfg, L.build_lut(L.PTF_PQ, 10, 1000.0, 0.01))
    assert m.used_rccl() == (ndev > 1), m.transport_note()     # one distinct device: nothing to broadcast, host copies
    if ndev == 1:
        m.set_transport(1)                                      # ... and through a one-rank RCCL communicator on request
        m.set_quantizer(*cfg, L.build_lut(L.PTF_PQ, 10, 1000.0, 0.01))
        assert m.used_rccl()
    orc = o.Oracle(*cfg)
    frames = [o.synth_frame(128, 64, 9, i) for i in range(7)]
    planes, st, means = m.encode_frames(frames, 20.0, 2)
    for f, pl in zip(frames, planes):
        e, _, _ = orc.encode(f.copy(), 20.0, 2)
        assert all(np.array_equal(a, b) for a, b in zip(pl, e))
    dec = m.decode_frames(planes, st, 128, 64, 20.0, 2)
    for pl, d in zip(planes, dec):
        assert np.array_equal(d.view(np.uint32), orc.decode(pl, st, 128, 64, 20.0, 2).view(np.uint32))
    # a second table through the same communicators
    m.set_quantizer(L.PTF_LOG, 12, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_LOG, 12))
    orc2 = o.Oracle(o.PTF_LOG, 12, o.CS_LUV, 8, 1e4, 0.005)
    planes2, _, _ = m.encode_frames(frames[:3], 1.0, 2)
    for f, pl in zip(frames, planes2):
        assert all(np.array_equal(a, b) for a, b in zip(pl, orc2.encode(f.copy(), 1.0, 2)[0]))
    assert m.ctx(0).quantizer_info()["mode"] == 3
    m.close()


def test_lumaenc_batch_mode_writes_the_same_stream(tmp_path):
    """tools/lumaenc with LUMAENC_SHARDS (LumaBatchEncoder, what it uses by itself when several GPUs are visible) writes byte
    for byte the stream the one-frame-at-a-time LumaEncoder loop writes"""
    exe = os.path.join(ROOT, "lumahdrv_amd", "bin", "lumaenc")
    a, b = str(tmp_path / "loop.lhs"), str(tmp_path / "batch.lhs")
    base = [exe, "-i", "__test__", "-f", "1:7"]
    env = dict(os.environ)
    env.pop("LUMAENC_SHARDS", None)
    import torch
    if torch.cuda.device_count() == 1:
        r = subprocess.run(base + ["-o", a], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
    else:
        r = subprocess.run(base + ["-o", a], capture_output=True, text=True, env=dict(env, LUMAENC_SHARDS="1"), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run(base + ["-o", b], capture_output=True, text=True, env=dict(env, LUMAENC_SHARDS="3", LUMAENC_FRAMES_PER_SHARD="1"),
                       timeout=300)
    assert r.returncode == 0 and "3 shard(s)" in r.stderr and "7 frames encoded" in r.stderr, r.stderr[-2000:]
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.parametrize("profile", [2, 3])
def test_ycbcr_stream_tables_do_not_change_a_bit(oracle_mod, profile):
    """YCbCr with the per-stream tables (composite luminance-code records on encode, y table on decode: the default) against the
    same kernels evaluating every PQ function per pixel (lumahip_tune "ycbcr_tables" 0) and against the oracle; with per-frame
    statistics requested the encode side takes the per-pixel path by itself and must still agree."""
    import torch
    import lumahdrv_amd as L
    o = oracle_mod
    dev = torch.device("cuda:0")
    cfg = CFGS["pq10_ycbcr"]
    sc = cfg[6]
    w, h, B = 1280, 720, 3
    n3 = 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    src = torch.empty(B * n3, dtype=torch.float32, device=dev)
    res = {}
    for tables in (1, 0):
        c = L.Context(0)
        c.tune("ycbcr_tables", tables)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.set_quantizer(cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5], L.build_lut(cfg[0], cfg[1], cfg[4], cfg[5]))
        c.synth_frames_device(src.data_ptr(), n3, B, w, h, 21, 0)
        # a few extreme pixels: black, huge, NaN, inf
        s3 = src.view(B, 3, h, w)
        s3[0, :, 0, 0:4] = 0.0
        s3[0, :, 0, 4:8] = 3.0e38
        s3[0, 0, 0, 8] = float("nan")
        s3[0, 1, 0, 12] = float("inf")
        planes = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
        stats = torch.zeros(3 * B, dtype=torch.float32, device=dev)
        c.encode_frames_device(src.data_ptr(), n3, B, w, h, sc, profile, [p.data_ptr() for p in planes], st, psz)
        planes_s = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
        c.encode_frames_device(src.data_ptr(), n3, B, w, h, sc, profile, [p.data_ptr() for p in planes_s], st, psz, stats.data_ptr())
        out = torch.empty(B * n3, dtype=torch.float32, device=dev)
        c.decode_frames_device([p.data_ptr() for p in planes], st, psz, B, w, h, profile, sc, out.data_ptr(), n3)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(planes, planes_s))
        res[tables] = ([p.cpu().numpy() for p in planes], out.cpu().numpy(), src.cpu().numpy().copy())
        c.set_stream(None)
        c.close()
    assert all(np.array_equal(a, b) for a, b in zip(res[1][0], res[0][0]))
    assert np.array_equal(res[1][1].view(np.uint32), res[0][1].view(np.uint32))
    orc = o.Oracle(cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5])
    f0 = res[1][2][:n3].reshape(3, h, w).copy()
    e, _, _ = orc.encode(f0, sc, profile)
    got = [res[1][0][p][:psz[p]].reshape(hs[p], st[p]) for p in range(3)]
    assert all(np.array_equal(a, b) for a, b in zip(got, e))
    assert np.array_equal(res[1][1][:n3].reshape(3, h, w).view(np.uint32), orc.decode(e, st, w, h, sc, profile).view(np.uint32))


@pytest.mark.gpu
def test_scalar_host_forms_equal_the_array_kernels(oracle_mod):
    """LumaQuantizer::quantize / dequantize per value on the host (lumahip_quantize_value_host, the facade's per-value members)
    against the array kernels (lumahip_quantize_array_host) on the same inputs, every configuration, both kinds of channel."""
    import lumahdrv_amd as L
    from lumahdrv_amd import capi
    from tests.golden.make_golden import CONFIGS
    rng = np.random.default_rng(77)
    special = np.array([0.0, -0.0, 1e-45, 1e-10, 1e-4, 1.0, 0.5, 0.4999999, 65504.0, 1e4, 1e8, 3e38, np.inf, -np.inf, np.nan, -1.0,
                        -1e-30], dtype=np.float32)
    for name, cfg in CONFIGS.items():
        ptf, bits, cs, bitsC, mx, mn = cfg
        q = L.LumaQuantizer()
        q.setQuantizer(*cfg)
        lut = q.getMapping()
        v = np.concatenate([special, np.exp(rng.uniform(np.log(1e-6), np.log(1e5), 3000)).astype(np.float32),
                            lut[rng.integers(0, lut.size, 500)], rng.uniform(-0.1, 1.1, 500).astype(np.float32)])
        for ch in (0, 1):
            arr = q.ctx.quantize_array(v, ch)
            sca = np.array([capi.quantize_value(lut, cs, bitsC, float(x), ch) for x in v], dtype=np.float32)
            assert np.array_equal(arr, sca), (name, ch, v[arr != sca][:5])
            codes = np.concatenate([np.arange(-3, (1 << (bits if ch == 0 else bitsC)) + 3, dtype=np.float32)[:5000],
                                    np.array([0.5, 1.999, np.nan, np.inf, -np.inf], dtype=np.float32)])
            arr = q.ctx.dequantize_array(codes, ch)
            sca = np.array([capi.quantize_value(lut, cs, bitsC, float(x), ch, dequantize=True) for x in codes], dtype=np.float32)
            assert np.array_equal(arr.view(np.uint32), sca.view(np.uint32)) , (name, ch)


@pytest.mark.gpu
def test_sync_and_host_calls_inside_an_unordered_section(oracle_mod):
    """lumahip_sync inside an open section waits for the section's lanes (results are read right after it, with the section
    still open), and entry points other than the four _device encode / decode calls -- here the host entry points with their
    own upload / kernel / download streams, and the stream push / pop -- keep to their own streams inside a section: a host call
    between two lane launches reads its frame after it is uploaded and returns the right planes."""
    import lumahdrv_amd as L
    o = oracle_mod
    cfg = (L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005)
    lut = L.build_lut(cfg[0], cfg[1], cfg[4], cfg[5])
    c = L.Context(0)                      # its own stream: nothing but lumahip_sync orders the host reads below
    c.set_quantizer(*cfg, lut)
    orc = o.Oracle(*cfg)
    w, h, B, NB, profile = 1280, 720, 4, 6, 2
    n3 = 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    d_src = c.malloc(NB * B * n3 * 4)
    d_pl = [c.malloc(NB * B * psz[p]) for p in range(3)]
    c.synth_frames_device(d_src, n3, NB * B, w, h, 9, 0)
    c.sync()
    hostf = o.synth_frame(w, h, 9, 1000)
    exp_host, _, _ = orc.encode(hostf.copy(), 1.0, profile)
    for rep in range(3):
        for p in range(3):
            c.h2d(d_pl[p], np.zeros(NB * B * psz[p], dtype=np.uint8))
        c.begin_unordered(2)
        for b in range(NB):
            c.encode_frames_device(d_src + b * B * n3 * 4, n3, B, w, h, 1.0, profile, [d_pl[p] + b * B * psz[p] for p in range(3)], st, psz)
            if b == 2:
                got, _, _ = c.encode_frame(hostf, 1.0, profile)            # a host call in the middle of the section
                for p in range(3):
                    assert np.array_equal(got[p], exp_host[p]), (rep, p)
        c.sync()                                                           # section still open
        planes = [np.empty(NB * B * psz[p], dtype=np.uint8) for p in range(3)]
        for p in range(3):
            rc = c.L.lumahip_memcpy_d2h(c.h, planes[p].ctypes.data, d_pl[p], planes[p].nbytes)
            assert rc == 0
        c.end_unordered()
        for f in (0, B * NB // 2, B * NB - 1):
            e, _, _ = orc.encode(o.synth_frame(w, h, 9, f), 1.0, profile)
            for p in range(3):
                assert np.array_equal(planes[p][f * psz[p]:(f + 1) * psz[p]].reshape(hs[p], st[p]), e[p]), (rep, f, p)
    c.free(d_src)
    for p in d_pl:
        c.free(p)
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_numa_placement_modes_do_not_change_results(oracle_mod, mode):
    """lumahip_tune numa 0 / 1 / 2 / 3 (nothing / rings + pinned copy threads / rings only, the default / threads only): where
    the pinned staging rings live and where the copy threads run changes no byte of a host-fed batch; lumahip_numa_info reports
    a node of this host (or -1 on a one-node host) and lumahip_numa_pin_current_thread leaves the caller inside that node."""
    import lumahdrv_amd as L
    o = oracle_mod
    cfg = (L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005)
    c = L.Context(0)
    c.tune("numa", mode)
    c.set_quantizer(*cfg, L.build_lut(cfg[0], cfg[1], cfg[4], cfg[5]))
    info = c.numa_info()
    nodes = [d for d in os.listdir("/sys/devices/system/node")] if os.path.isdir("/sys/devices/system/node") else []
    nnodes = len([d for d in nodes if d.startswith("node") and d[4:].isdigit()])
    if mode == 0 or nnodes < 2:
        assert info["node"] == -1 and info["cpus"] == 0
    else:
        assert 0 <= info["node"] < nnodes and info["cpus"] > 0
    orc = o.Oracle(*cfg)
    frames = [o.synth_frame(1920, 1080, 3, i) for i in range(5)]          # large enough for the copy threads to be used
    planes, st, _ = c.encode_frames(frames, 1.0, 2)
    for f, pl in zip(frames, planes):
        e, _, _ = orc.encode(f.copy(), 1.0, 2, threads=4)
        assert all(np.array_equal(a, b) for a, b in zip(pl, e))
    dec = c.decode_frames(planes, st, 1920, 1080, 1.0, 2)
    assert np.array_equal(dec[0].view(np.uint32), orc.decode(planes[0], st, 1920, 1080, 1.0, 2, threads=4).view(np.uint32))
    before = os.sched_getaffinity(0)
    assert c.L.lumahip_numa_pin_current_thread(c.h) == 0
    after = os.sched_getaffinity(0)
    if mode in (1, 3) and info["node"] >= 0:
        assert len(after) == info["cpus"] and after <= before
    else:
        assert after == before
    os.sched_setaffinity(0, before)
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sc", [1.0, 20.0, 65536.0, 65537.0, 1e6, 1.0 / 65536.0, 1e-6, 3e38])
def test_ycbcr_decode_prescalings_inside_and_outside_the_short_division_range(oracle_mod, sc):
    """The YCbCr decode kernels divide by preScaling with a 5-operation quotient when sc is in [2^-16, 2^16] (two copies of the
    unit's code, chosen per launch) and send everything else -- and every value the exponent guard turns away -- to the complete
    functions with IEEE division: decoded floats 0 ulp against the oracle on both sides of both limits, random codes plus the
    darkest ones (where the decoded value underflows the guard)."""
    import lumahdrv_amd as L
    o = oracle_mod
    cfg = (L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01)
    q = L.LumaQuantizer()
    q.setQuantizer(*cfg)
    orc = o.Oracle(*cfg)
    rng = np.random.default_rng(int(np.float32(sc).view(np.uint32)))
    for profile, (w, h) in ((2, (256, 64)), (3, (128, 32)), (0, (64, 16))):
        _, hs, st, bps = L.plane_geometry(w, h, profile)
        ws = (w, w // 2 if profile in (0, 2) else w, w // 2 if profile in (0, 2) else w)
        planes = []
        for p in range(3):
            hi = 256 if bps == 1 else 1030
            codes = rng.integers(0, hi, size=(hs[p], ws[p]))
            if p == 0:
                codes[0, :] = rng.integers(0, 6, size=ws[p])        # the darkest luminance codes
            else:
                codes[:2, :] = rng.integers(500, 530, size=(2, ws[p]))   # near-neutral chroma: sums that cancel
            buf = np.zeros((hs[p], st[p]), dtype=np.uint8)
            if bps == 2:
                buf[:, :2 * ws[p]] = codes.astype("<u2").view(np.uint8).reshape(hs[p], 2 * ws[p])
            else:
                buf[:, :ws[p]] = codes.astype(np.uint8)
            planes.append(buf)
        got = q.ctx.decode_frame(planes, st, w, h, sc, profile)
        with np.errstate(all="ignore"):
            exp = orc.decode(planes, st, w, h, sc, profile)
        same = (got.view(np.uint32) == exp.view(np.uint32)) | (np.isnan(got) & np.isnan(exp))
        assert bool(same.all()), (sc, profile, np.argwhere(~same)[:4].tolist())

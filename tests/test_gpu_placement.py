"""lumahdrv_amd.placement.HbmChunkPool on a real MI355X: the pool builds, reports its groups, hands out chunks, and frames
encoded into chunk-placed buffers give the bytes they give in plainly allocated buffers (placement never touches results)."""
import numpy as np
import pytest

from lumahdrv_amd.placement import as_tensor as placement_tensor

pytestmark = pytest.mark.gpu


def test_chunk_pool_builds_and_placement_does_not_change_results(oracle_mod):
    import torch
    import lumahdrv_amd as L
    from lumahdrv_amd.placement import CHUNK_BYTES, HbmChunkPool
    dev = torch.device("cuda:0")
    torch.cuda.empty_cache()
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    pool = HbmChunkPool(ctx, dev, n_float=3, n_y=1, n_uv=1, n_striped=2)
    st_ = pool.stats
    assert st_["chunks"] >= 5 and len(pool.float) == 3 and len(pool.y) == 1 and len(pool.uv) == 1
    assert [len(g) for g in pool.striped] == [2, 2, 2]
    if st_["grouped"]:      # the striped chunks of list g really lie in region group g
        assert all(pool.group_of(t) == g for g in range(3) for t in pool.striped[g])
    if st_["grouped"]:
        assert sum(st_["groups"]) == st_["chunks"] and len(st_["groups"]) >= 2
        pm = st_["probe_ms"]
        # the point of the exercise: the chosen layout is not slower than input and planes sharing one group
        assert pm["float_chunks_kept_slowest"] <= pm["input_and_planes_in_one_group"] * 1.01
    free_after, _ = torch.cuda.mem_get_info(dev)
    assert free_after > 100 * 2 ** 30        # everything that was not kept went back to the driver

    w, h, B, profile = 1280, 720, 4, 2
    n3 = 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    src_c, = pool.take_float(1)
    uv_c, = pool.take_uv(1)
    y_c, = pool.take_y(1)
    for c in (uv_c, y_c):
        c.zero_()
    ctx.synth_frames_device(src_c.data_ptr(), n3, B, w, h, 7, 0)
    pl = [y_c.data_ptr(), uv_c.data_ptr(), uv_c.data_ptr() + (1 << 28)]
    ctx.encode_frames_device(src_c.data_ptr(), n3, B, w, h, 1.0, profile, pl, st, psz)
    # the same frames through plain tensors
    src = torch.empty(B * n3, dtype=torch.float32, device=dev)
    planes = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    ctx.synth_frames_device(src.data_ptr(), n3, B, w, h, 7, 0)
    ctx.encode_frames_device(src.data_ptr(), n3, B, w, h, 1.0, profile, [p.data_ptr() for p in planes], st, psz)
    torch.cuda.synchronize()
    assert torch.equal(src_c[:B * n3 * 4].view(torch.float32), src)
    assert torch.equal(y_c[:B * psz[0]], planes[0])
    assert torch.equal(uv_c[:B * psz[1]], planes[1])
    assert torch.equal(uv_c[(1 << 28):(1 << 28) + B * psz[2]], planes[2])
    # and against the oracle for the first frame
    o = oracle_mod
    orc = o.Oracle(o.PTF_PQ, 11, o.CS_LUV, 8, 1e4, 0.005)
    e, _, _ = orc.encode(o.synth_frame(w, h, 7, 0), 1.0, profile)
    got_y = y_c[:psz[0]].cpu().numpy().reshape(hs[0], st[0])
    assert np.array_equal(got_y, e[0])
    pool.give_back([src_c], [y_c], [uv_c])
    pool.close()
    ctx.set_stream(None)
    ctx.close()
    assert CHUNK_BYTES == 2 << 30


def test_pool_stream_cpp_example():
    """tools/pool_stream.cpp: a resident stream carved from lumahip_pool_* from plain C++ over the C ABI (no Python, no torch),
    against the same stream in lumahip_malloc buffers: same bytes, and -- where the box shows region groups -- not slower."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "lumahdrv_amd", "bin", "pool_stream")
    r = subprocess.run([exe, "6", "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    # (no rate against rate: which of the two is faster, and by how much, is reported in profiles/, not asserted on a shared box)
    assert d["planes_identical"] is True and d["batches"] == 6 and d["pooled_mpix_s"] > 0 and d["plain_mpix_s"] > 0
    # the decode half: packed LumaFrames in chunks the pool handed out in its ROTATING mode, same floats as in plain buffers
    assert d["decoded_identical"] is True and d["decode_packed_rotating_mpix_s"] > 0 and d["decode_packed_plain_mpix_s"] > 0
    if d["pool"].get("grouped"):
        assert d["decode_ring_groups"] == [0, 1, 2, 0, 1, 2]


def test_rotating_allocations_walk_the_region_groups():
    """lumahip_pool_alloc(LUMAHIP_POOL_ROTATING): consecutive allocations hand out the striped chunks of groups 0, 1, 2, 0, ... (a
    caller that allocates the packed output buffer of every batch in stream order gets launches in flight that write different
    groups); a group that has run out is skipped; decoding into such chunks gives the bytes a plain buffer gets."""
    import torch
    import lumahdrv_amd as L
    from lumahdrv_amd import capi
    from lumahdrv_amd.placement import HbmChunkPool
    dev = torch.device("cuda:0")
    torch.cuda.empty_cache()
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    pool = HbmChunkPool(ctx, dev, n_float=1, n_y=1, n_uv=1, n_striped=2)
    grouped = pool.stats["grouped"]
    assert pool.pool.available(capi.POOL_ROTATING) == 0            # (the wrapper holds every striped chunk)
    ring = pool.take_rotating(5)
    if grouped:
        assert [pool.group_of(t) for t in ring] == [0, 1, 2, 0, 1]
    assert sorted(len(g) for g in pool.striped) == [0, 0, 1]       # one chunk left, in the group the walk would visit next
    with pytest.raises(RuntimeError):
        pool.take_rotating(2)
    last, = pool.take_rotating(1)                                  # groups 0 and 1 have run out: the walk skips to what is left
    if grouped:
        assert pool.group_of(last) == 2
    w, h, B = 1280, 720, 3
    n3 = 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    psz = [hs[p] * st[p] for p in range(3)]
    src = torch.empty(B * n3, dtype=torch.float32, device=dev)
    planes = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    ref = torch.empty(B * n3, dtype=torch.float32, device=dev)
    ctx.synth_frames_device(src.data_ptr(), n3, B, w, h, 3, 0)
    pl = [p.data_ptr() for p in planes]
    ctx.encode_frames_device(src.data_ptr(), n3, B, w, h, 1.0, 2, pl, st, psz)
    ctx.decode_frames_device(pl, st, psz, B, w, h, 2, 1.0, ref.data_ptr(), n3)
    ctx.begin_unordered(2)
    for t in ring:
        ctx.decode_frames_device(pl, st, psz, B, w, h, 2, 1.0, t.data_ptr(), n3)
    ctx.end_unordered()
    torch.cuda.synchronize()
    for t in ring:
        assert torch.equal(t[:B * n3 * 4].view(torch.int32), ref.view(torch.int32))
    pool.give_back_rotating(ring + [last])
    assert [len(g) for g in pool.striped] == [2, 2, 2] or not grouped
    pool.close()
    ctx.set_stream(None)
    ctx.close()


def test_packed_frames_rotating_over_three_buffers_equal_the_packed_batch(oracle_mod):
    """lumahip_decode_frames_device_rotating: frame f of the batch at bases[f % 3] + (f // 3) * frame_stride, tiles interleaved over the
    frames.  Every frame must hold the floats lumahip_decode_frames_device writes for it -- Lu'v' and YCbCr (its own kernels), 4:2:0
    and 4:4:4, widths with four and two pixels per thread, batches of 1, 2, 3 and 7 frames (not multiples of three) -- and frame 0
    the oracle's."""
    import torch
    import lumahdrv_amd as L
    o = oracle_mod
    dev = torch.device("cuda:0")
    for cfg, sc in (((L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005), 1.0), ((L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01), 20.0)):
        ctx = L.Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_quantizer(*cfg, L.build_lut(cfg[0], cfg[1], cfg[4], cfg[5]))
        orc = o.Oracle(*cfg)
        for (w, h), profile, B in (((640, 96), 2, 7), ((258, 34), 2, 3), ((256, 64), 3, 2), ((64, 32), 2, 1)):
            n1, n3 = w * h, 3 * w * h
            _, hs, st, _ = L.plane_geometry(w, h, profile)
            psz = [hs[p] * st[p] for p in range(3)]
            src = torch.empty(B * n3, dtype=torch.float32, device=dev)
            planes = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
            ctx.synth_frames_device(src.data_ptr(), n3, B, w, h, 11, 0)
            pl = [p.data_ptr() for p in planes]
            ctx.encode_frames_device(src.data_ptr(), n3, B, w, h, sc, profile, pl, st, psz)
            ref = torch.empty(B * n3, dtype=torch.float32, device=dev)
            ctx.decode_frames_device(pl, st, psz, B, w, h, profile, sc, ref.data_ptr(), n3)
            per = -(-B // 3)
            fs = n3 + 64                                              # frames of one buffer a little apart
            bufs = [torch.full((per * fs,), float("nan"), dtype=torch.float32, device=dev) for _ in range(3)]
            ctx.decode_frames_device_rotating(pl, st, psz, B, w, h, profile, sc, [b.data_ptr() for b in bufs], fs)
            torch.cuda.synchronize()
            for f in range(B):
                got = bufs[f % 3][(f // 3) * fs:(f // 3) * fs + n3]
                assert torch.equal(got.view(torch.int32), ref[f * n3:(f + 1) * n3].view(torch.int32)), (cfg[2], (w, h), profile, B, f)
            for b_ in bufs:                                           # nothing written between or behind the frames
                gaps = b_.view(per, fs)[:, n3:]
                assert bool(torch.isnan(gaps).all())
            pf = [planes[p][:psz[p]].cpu().numpy().reshape(hs[p], st[p]) for p in range(3)]
            exp = orc.decode(pf, st, w, h, sc, profile)
            assert np.array_equal(bufs[0][:n3].cpu().numpy().view(np.uint32), exp.reshape(-1).view(np.uint32))
        with pytest.raises(L.LumaHipError):                           # one buffer twice
            ctx.decode_frames_device_rotating(pl, st, psz, B, w, h, profile, sc, [bufs[0].data_ptr(), bufs[0].data_ptr(), bufs[2].data_ptr()], n3)
        # buffers whose extents over the batch overlap, and a second / third buffer the vector stores cannot take: rejected, nothing written
        w, h, profile, B = 258, 34, 2, 7                              # (two pixels per thread: 8-byte stores)
        n3 = 3 * w * h
        _, hs, st, _ = L.plane_geometry(w, h, profile)
        psz = [hs[p] * st[p] for p in range(3)]
        planes = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
        pl = [p.data_ptr() for p in planes]
        big = torch.full((12 * n3,), float("nan"), dtype=torch.float32, device=dev)
        base = big.data_ptr()
        ok = [base, base + 3 * n3 * 4, base + 6 * n3 * 4]             # three frames each: exactly what B = 7 needs of buffer 0
        ctx.decode_frames_device_rotating(pl, st, psz, B, w, h, profile, sc, ok, n3)
        torch.cuda.synchronize()
        big.fill_(float("nan"))
        for bad in ([base, base + 2 * n3 * 4, base + 6 * n3 * 4],     # buffer 1 starts inside buffer 0's third frame
                    [base, base + 3 * n3 * 4, base + 4 * n3 * 4 + 8], # buffer 2 inside buffer 1
                    [base, base + 3 * n3 * 4 + 4, base + 6 * n3 * 4], # buffer 1 only 4-byte aligned
                    [base, base + 3 * n3 * 4, base + 6 * n3 * 4 + 4]):
            with pytest.raises(L.LumaHipError):
                ctx.decode_frames_device_rotating(pl, st, psz, B, w, h, profile, sc, bad, n3)
        torch.cuda.synchronize()
        assert bool(torch.isnan(big).all())
        ctx.set_stream(None)
        ctx.close()


def test_library_allocated_decoded_ring_holds_the_packed_batch(oracle_mod):
    """lumahip_decoded_ring_*: the caller lets the library allocate the decoded frames.  Every frame pointer it hands out holds, after
    lumahip_decode_frames_device_ring, the packed LumaFrame lumahip_decode_frames_device writes for that frame -- Lu'v' and YCbCr,
    full and short batches, every slot of the ring -- and frame 0 the oracle's; frames of a batch do not overlap; bad arguments
    are refused.  Where the ring lives (region groups found or plain allocations) never changes a float."""
    import torch
    import lumahdrv_amd as L
    from lumahdrv_amd import capi
    o = oracle_mod
    dev = torch.device("cuda:0")
    for cfg, sc in (((L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005), 1.0), ((L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01), 20.0)):
        ctx = L.Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_quantizer(*cfg, L.build_lut(cfg[0], cfg[1], cfg[4], cfg[5]))
        orc = o.Oracle(*cfg)
        for (w, h), profile, B, slots in (((640, 96), 2, 7, 3), ((258, 34), 2, 4, 2), ((64, 32), 3, 1, 1)):
            n3 = 3 * w * h
            _, hs, st, _ = L.plane_geometry(w, h, profile)
            psz = [hs[p] * st[p] for p in range(3)]
            src = torch.empty(B * n3, dtype=torch.float32, device=dev)
            planes = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
            ctx.synth_frames_device(src.data_ptr(), n3, B, w, h, 13, 0)
            pl = [p.data_ptr() for p in planes]
            ctx.encode_frames_device(src.data_ptr(), n3, B, w, h, sc, profile, pl, st, psz)
            ref = torch.empty(B * n3, dtype=torch.float32, device=dev)
            ctx.decode_frames_device(pl, st, psz, B, w, h, profile, sc, ref.data_ptr(), n3)
            ring = capi.DecodedRing(ctx, slots, B, w, h)
            assert ring.nbatches == slots and ring.nframes == B and ring.frame_stride == n3
            for slot in range(slots):
                nf = B if slot % 2 == 0 else max(1, B - 2)            # a short batch into a full slot
                ring.decode(pl, st, psz, nf, profile, sc, slot)
                torch.cuda.synchronize()
                spans = []
                for f in range(nf):
                    p = ring.frame_ptr(slot, f)
                    assert p % 16 == 0
                    got = placement_tensor(p, n3 * 4, dev).view(torch.int32)
                    assert torch.equal(got, ref[f * n3:(f + 1) * n3].view(torch.int32)), (cfg[2], (w, h), profile, slot, f)
                    spans.append((p, p + n3 * 4))
                spans.sort()
                assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
            pf = [planes[p][:psz[p]].cpu().numpy().reshape(hs[p], st[p]) for p in range(3)]
            exp = orc.decode(pf, st, w, h, sc, profile)
            got0 = placement_tensor(ring.frame_ptr(0, 0), n3 * 4, dev).view(torch.float32).cpu().numpy()
            assert np.array_equal(got0.view(np.uint32), exp.reshape(-1).view(np.uint32))
            with pytest.raises(L.LumaHipError):
                ring.frame_ptr(slots, 0)
            with pytest.raises(L.LumaHipError):
                ring.frame_ptr(0, B)
            with pytest.raises(L.LumaHipError):
                ring.decode(pl, st, psz, B + 1, profile, sc, 0)       # more frames than a slot holds
            with pytest.raises(L.LumaHipError):
                ring.decode(pl, st, psz, B, profile, sc, slots)
            ring.close()
        ctx.set_stream(None)
        ctx.close()


def test_small_pool_finds_groups_among_a_dozen_chunks_and_changes_no_byte(oracle_mod):
    """lumahip_pool_create_small (round 6): the pool a caller that shares the GPU can afford -- it may take at most a dozen chunks for
    its probes (not all free memory), keeps the 2 + 1 + 1 asked for and returns the rest; planes encoded into its chunks are the
    bytes plain buffers get and frame 0 the oracle's.  Whether groups were found is reported, never assumed (a box without the
    effect gives an ungrouped pool and the same bytes)."""
    import torch
    import lumahdrv_amd as L
    from lumahdrv_amd.placement import CHUNK_BYTES, HbmChunkPool
    o = oracle_mod
    dev = torch.device("cuda:0")
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info(dev)[0]
    cfg = (L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005)
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(*cfg, L.build_lut(cfg[0], cfg[1], cfg[4], cfg[5]))
    pool = HbmChunkPool(ctx, dev, 2, 1, 1, 0, small=True)
    st_ = pool.stats
    assert len(pool.float) == 2 and len(pool.y) == 1 and len(pool.uv) == 1
    assert 4 <= st_["chunks"] <= 16                                   # probed: a dozen, not the ~140 the full pool takes
    assert free0 - torch.cuda.mem_get_info(dev)[0] <= 5 * CHUNK_BYTES   # kept: the four chunks (+ allocator slack)
    if st_.get("grouped"):
        assert len(st_["groups"]) >= 2 and pool.group_of(pool.y[0]) != pool.group_of(pool.float[0])
    w, h, B = 1920, 1080, 3
    n3 = 3 * w * h
    _, hs, stp, _ = L.plane_geometry(w, h, 2)
    psz = [hs[p] * stp[p] for p in range(3)]
    src = pool.float[0]
    ctx.synth_frames_device(src.data_ptr(), n3, B, w, h, 21, 0)
    placed = [pool.y[0].data_ptr(), pool.uv[0].data_ptr(), pool.uv[0].data_ptr() + (64 << 20)]
    plain = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    ctx.encode_frames_device(src.data_ptr(), n3, B, w, h, 1.0, 2, placed, stp, psz)
    ctx.encode_frames_device(src.data_ptr(), n3, B, w, h, 1.0, 2, [t.data_ptr() for t in plain], stp, psz)
    torch.cuda.synchronize()
    for p in range(3):
        got = placement_tensor(placed[p], B * psz[p], dev)
        assert torch.equal(got, plain[p]), p
    f0 = src[:n3 * 4].view(torch.float32).cpu().numpy().reshape(3, h, w)
    exp, _, _ = o.Oracle(*cfg).encode(f0.copy(), 1.0, 2)
    for p in range(3):
        assert np.array_equal(plain[p][:psz[p]].cpu().numpy().reshape(hs[p], stp[p]), exp[p])
    pool.close()
    ctx.set_stream(None)
    ctx.close()

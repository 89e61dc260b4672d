"""The N>1 path on CPU: world_size-2 gloo.  Sharding plan, quantizer broadcast and in-order reassembly are the
product's code (lumahdrv_amd/sharding.py); the per-frame work is injected -- here the oracle stands in for the
GPU (tests may use it), so the digests must equal a single-process run over the whole stream."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_stream():
    from lumahdrv_amd.sharding import owner_of, shard_range
    for n in (0, 1, 7, 8, 9, 250, 2000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                rg = shard_range(n, r, world)
                seen.extend(rg)
                assert all(owner_of(f, n, world) == r for f in rg)
            assert seen == list(range(n))            # contiguous, in stream order, nothing lost or doubled
    assert [len(shard_range(2000, r, 8)) for r in range(8)] == [250] * 8   # BASELINE configs[4]
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nframes, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lumahdrv_amd import capi
    from lumahdrv_amd.sharding import ShardedStream
    from oracle import oracle_py as o
    cfg = lut = None
    if rank == 0:   # only rank 0 knows the configuration and builds the table
        cfg = (capi.PTF_PQ, 11, capi.CS_LUV, 8, 1e4, 0.005, 1.0, 2)
        lut = capi.build_lut(capi.PTF_PQ, 11, 1e4, 0.005)
    st = ShardedStream(nframes, torch.device("cpu"), cfg, lut)

    def make_worker(cfg, lut):
        orc = o.Oracle(cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5])
        orc.overwrite_mapping(lut)            # the broadcast table, not a locally rebuilt one
        return orc

    def process(orc, f):
        planes, _, _ = orc.encode(o.synth_frame(64, 32, frame=f), st.cfg[6], st.cfg[7])
        return o.fnv1a64(np.concatenate([p.ravel() for p in planes]))

    digests = st.run(make_worker, process)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array(digests, dtype=np.int64))
    np.save(os.path.join(out_dir, "lut%d.npy" % rank), st.lut)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_stream_matches_single_process(tmp_path):
    from oracle import oracle_py as o
    nframes, world = 7, 2
    mp.spawn(_worker, args=(world, _free_port(), nframes, str(tmp_path)), nprocs=world, join=True)
    orc = o.Oracle(o.PTF_PQ, 11, o.CS_LUV, 8, 1e4, 0.005)
    expect = []
    for f in range(nframes):
        planes, _, _ = orc.encode(o.synth_frame(64, 32, frame=f), 1.0, 2)
        expect.append(o.fnv1a64(np.concatenate([p.ravel() for p in planes])) & 0x7FFFFFFFFFFFFFFF)
    for r in range(world):
        assert np.load(tmp_path / ("rank%d.npy" % r)).tolist() == expect
        assert np.array_equal(np.load(tmp_path / ("lut%d.npy" % r)).view(np.uint32), orc.mapping.view(np.uint32))

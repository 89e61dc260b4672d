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


def test_c_abi_shard_range_equals_the_python_plan():
    """lumahip_shard_range (the many-GPU layer of the C ABI) and lumahdrv_amd.sharding.shard_range (the one-process-per-GPU
    path) must cut a stream identically, so that either driver gives every GPU the same block"""
    from lumahdrv_amd import capi
    from lumahdrv_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 9, 40, 250, 2000):
        for world in (1, 2, 3, 5, 8):
            for r in range(world):
                assert capi.shard_range(n, r, world) == shard_range(n, r, world)
    with pytest.raises(capi.LumaHipError):
        capi.shard_range(10, 2, 2)


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_bench_respawn_command_line():
    """`python bench.py --gpus N` re-launches itself as the driver would launch it: one rank per GPU of one node, rendezvous
    on 127.0.0.1, its own arguments passed through unchanged"""
    b = _bench_module()
    cmd = b.respawn_command(4, ["--gpus", "4", "--steps", "7", "--warmup", "2"], 29611)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29611"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]


def _timer_worker(rank, world, port, out_dir):
    import json
    import sys
    import time
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = _bench_module()
    calls = []

    def step(i):
        calls.append(i)
        time.sleep(0.004 if rank == 1 else 0.001)     # rank 1 is the slow one

    # rank 0 alone would stop after ~5 regions (5 x 3 x 1 ms... its own device time), rank 1 later: rank 0 decides for both
    tm = b.Timer(3, 2, True, torch.device("cpu"), 0.02, 50)
    res = tm.run(step)
    json.dump({"repeats": res["repeats"], "calls": calls, "wall_median": res["wall_median"], "ranks": res["rank_wall_medians"]},
              open(os.path.join(out_dir, "timer%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_timer_loop_with_two_ranks(tmp_path):
    """bench.py's Timer under world 2 (gloo, CPU): both ranks run the same number of regions (rank 0 decides), exactly K steps
    per region after W warm-up steps, the reported region time is the slowest rank's, and the per-rank medians show which
    rank that is"""
    import json
    world = 2
    mp.spawn(_timer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [json.load(open(tmp_path / ("timer%d.json" % k))) for k in range(world)]
    assert r[0]["repeats"] == r[1]["repeats"] >= 1
    n = 2 + 3 * r[0]["repeats"]                                    # W + K x regions
    assert r[0]["calls"] == r[1]["calls"] == list(range(n))
    assert r[0]["wall_median"] == pytest.approx(r[1]["wall_median"])        # MAX over ranks: the same number everywhere
    assert r[0]["ranks"] == pytest.approx(r[1]["ranks"]) and len(r[0]["ranks"]) == 2
    assert r[0]["ranks"][1] > r[0]["ranks"][0]                     # rank 1 slept longer
    assert r[0]["wall_median"] >= 0.999 * max(r[0]["ranks"]) - 1e-3

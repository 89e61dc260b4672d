"""The host logic of the HBM chunk pool -- lumahip_pool_find_groups of the C ABI (lumahdrv_amd/csrc/lumahip_pool.hip) through
lumahdrv_amd/placement.py, and the slot arithmetic -- on synthetic timings.  CPU only; the pool itself is exercised on the
GPU by tests/test_gpu_placement.py."""
import random

from lumahdrv_amd.placement import CHUNK_BYTES, find_groups, plane_slots, slots


def _probe_for(group_of, fast=0.428, slow=0.457, noise=0.002, seed=1):
    rng = random.Random(seed)
    calls = []

    def probe(i, r):
        calls.append((i, r))
        base = slow if group_of[i] == group_of[r] else fast
        return base * (1.0 + rng.uniform(-noise, noise))
    return probe, calls


def test_find_groups_recovers_the_partition():
    # the layout measured on one box: runs of regions of three groups, interleaved
    group_of = [2] + [0] * 10 + [1] * 2 + [0] * 2 + [2] * 13 + [1] * 7 + [0] * 5
    probe, calls = _probe_for(group_of)
    groups, fast, probes = find_groups(len(group_of), probe)
    assert probes == len(calls) and probes < 3 * len(group_of)
    assert sorted(sorted(g) for g in groups) == sorted(sorted(i for i, g in enumerate(group_of) if g == k) for k in (0, 1, 2))
    assert 0.42 < fast < 0.435


def test_find_groups_without_contrast_reports_none():
    probe, _ = _probe_for([0] * 12)
    groups, _, _ = find_groups(12, probe)
    assert groups is None
    groups, _, _ = find_groups(1, probe)
    assert groups == [[0]]


def test_slots():
    # 4K, 20 frames per batch: Y 331.8 MB, U + V 165.9 MB
    ypc, yslot = slots(CHUNK_BYTES, 20 * 2160 * 7680)
    assert ypc == 6 and yslot % (64 << 20) == 0 and yslot >= 20 * 2160 * 7680
    uvpc, _ = slots(CHUNK_BYTES, 2 * 20 * 1080 * 3840 + (1 << 20))
    assert uvpc == 10 or uvpc == 12
    per_chunk, slot, offs = plane_slots(CHUNK_BYTES, [20 * 2160 * 7680, 20 * 1080 * 3840, 20 * 1080 * 3840])
    assert per_chunk == 4 and offs[0] == 0 and offs[1] >= 20 * 2160 * 7680 and offs[2] >= offs[1] + 20 * 1080 * 3840
    assert slots(1 << 20, 2 << 20)[0] == 0

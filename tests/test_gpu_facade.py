"""The C++ facade (include/luma/*.h over liblumahip.so) driven like the reference's test_simple_enc /
test_simple_dec, checked against the oracle.  The compile + link step also runs on CPU (test_host_side.py)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_facade_test(tmp):
    exe = os.path.join(tmp, "facade_roundtrip")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "facade_roundtrip.cpp"), "-o", exe,
                    "-L" + os.path.join(ROOT, "lumahdrv_amd", "lib"), "-lluma_hip", "-llumahip",
                    "-Wl,-rpath," + os.path.join(ROOT, "lumahdrv_amd", "lib")], check=True)
    return exe


@pytest.mark.gpu
@pytest.mark.parametrize("cs,bits,profile", [(0, 11, 2), (2, 10, 2), (3, 12, 3), (0, 8, 0)])
def test_facade_roundtrip_matches_oracle(oracle_mod, tmp_path, cs, bits, profile):
    o = oracle_mod
    exe = build_facade_test(str(tmp_path))
    w, h = 320, 180
    stream = str(tmp_path / "out.lhs")
    r = subprocess.run([exe, stream, str(w), str(h), "3", str(cs), str(bits), str(profile)], capture_output=True,
                       text=True, check=True)
    got = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines() if " " in l)
    orc = o.Oracle(o.PTF_PQ, bits, cs, 8, 1e4, 0.005)
    f = o.test_frame(w, h)
    assert got["input"] == "%016x" % o.fnv1a64(f)
    planes, st, avg = orc.encode(f, 1.0, profile)
    bps = 2 if profile > 1 else 1
    cw = w // 2 if profile in (0, 2) else w
    assert got["Y"] == "%016x" % o.fnv1a64(o.packed_rows(planes[0], bps * w))
    assert got["U"] == "%016x" % o.fnv1a64(o.packed_rows(planes[1], bps * cw))
    assert got["V"] == "%016x" % o.fnv1a64(o.packed_rows(planes[2], bps * cw))
    assert got["transformed"] == "%016x" % o.fnv1a64(f)        # in-place compat = the reference's side effect
    assert float(got["mean"]) == pytest.approx(avg, rel=1e-3)
    dec = orc.decode(planes, st, w, h, 1.0, profile)
    assert got["decoded"].split()[0] == "%016x" % o.fnv1a64(dec)
    assert "3 frames decoded" in r.stdout and "size %d" % ((1 << bits) - 1) in r.stdout
    assert got["odd-size:"] == "Invalid frame size"             # src/luma_encoder.cpp:118-119

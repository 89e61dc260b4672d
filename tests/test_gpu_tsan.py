"""The host plumbing of liblumahip.so under ThreadSanitizer -- needs an MI355X.

`make -C lumahdrv_amd/csrc tsan` builds lumahdrv_amd/lib_tsan/{liblumahip.so, libluma_hip.so} with the HOST side of every
translation unit instrumented (-Xarch_host -fsanitize=thread; the device code is compiled as always): the copy-thread pool
(generation counter + spin + condition variable), the staging rings and deferred downloads, the stream push / pop pipeline and
the per-shard threads of lumahip_multi_*.  The C++ tests that drive those paths hardest are linked against that build and run
with tests/tsan.supp (which names ROCm libraries only): a data race in this repository's code makes them exit 66."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"
LIB = os.path.join(ROOT, "lumahdrv_amd", "lib_tsan")


@pytest.fixture(scope="module")
def tsan_lib():
    if not os.path.exists(CLANGXX):
        pytest.skip("no ROCm clang++")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "lumahdrv_amd", "csrc"), "tsan", "-j", "8"], check=True)
    return LIB


def _build(tmp, name):
    exe = os.path.join(tmp, name + "_tsan")
    subprocess.run([CLANGXX, "-fsanitize=thread", "-g", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe, "-L" + LIB, "-lluma_hip", "-llumahip",
                    "-Wl,-rpath," + LIB], check=True)
    return exe


def _run(cmd, timeout=900):
    env = dict(os.environ)
    env["TSAN_OPTIONS"] = "suppressions=%s exitcode=66 halt_on_error=0 report_signal_unsafe=0 second_deadlock_stack=1" % os.path.join(ROOT, "tests", "tsan.supp")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    races = [l for l in r.stderr.splitlines() if "WARNING: ThreadSanitizer" in l]
    assert r.returncode == 0 and not races, "exit %d, %d TSan reports\n%s\n%s" % (r.returncode, len(races), r.stdout[-1500:], r.stderr[-6000:])
    return r.stdout


def test_multi_shard_stress_under_tsan(tsan_lib, tmp_path):
    out = _run([_build(str(tmp_path), "multi_stress"), "40", "8"])
    assert "OK multi_stress: 40 iterations, 8 shards" in out


def test_multi_batch_under_tsan(tsan_lib, tmp_path):
    out = _run([_build(str(tmp_path), "multi_batch"), "320", "180", "24", str(tmp_path / "b.lhs")])
    assert "OK all" in out and "OK LumaBatchEncoder" in out


def test_pipelined_encoder_and_decoder_under_tsan(tsan_lib, tmp_path):
    out = _run([_build(str(tmp_path), "pipelined_encoder"), str(tmp_path), "1920", "1080", "9", "0", "11", "2"])
    assert "OK streams identical: 9 frames" in out and "OK pipelined decode: 9 frames identical" in out

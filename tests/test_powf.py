"""lumahdrv_amd/csrc/pow_glibc.hpp (the restatement of glibc 2.35's powf that the YCbCr kernels run on the GPU)
against this host's libm powf, on the host: every 509th fp32 bit pattern of [0, +inf] plus a sweep of the
negative / NaN half, for the four exponents LumaQuantizer::transformPQ uses (src/luma_quantizer.cpp:485-501).
The exhaustive version (stride 1: 8.56e9 arguments, ~45 s on 8 cores, 0 mismatches on glibc 2.35-0ubuntu3.11)
is `tools/verify_powf.cpp` run by hand.  CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restated_powf_matches_host_libm(tmp_path):
    exe = str(tmp_path / "verify_powf")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-pthread", "-o", exe,
                    os.path.join(ROOT, "tools", "verify_powf.cpp"), "-lm"], check=True)
    r = subprocess.run([exe, "509"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout.splitlines()[-1]
    assert r.stdout.count("0 mismatches") == 7   # four exponents + the two ranges of the folded form (round 6) + the total
    assert r.stdout.count("folded form") == 2

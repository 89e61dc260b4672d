"""The oracle (oracle/luma_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer: every colour space x VP9
profile on a ragged frame with NaN / Inf / negative pixels, odd strides, multi-threaded bands.  SURVEY.md section 5
recommends exactly this for the CPU restatement, because the reference itself has latent UB (quirk 2: float ->
unsigned char of values > 255) that the restatement must state explicitly rather than inherit.  CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_sanitize")
    subprocess.run(["gcc", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-fno-omit-frame-pointer", "-o", exe, os.path.join(ROOT, "tests", "cpp", "oracle_sanitize.c"),
                    os.path.join(ROOT, "oracle", "luma_oracle.c"), "-lm", "-lpthread"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok ")

"""The CPU oracle's C code under AddressSanitizer + UndefinedBehaviorSanitizer (tests/oracle_sanitize.c): random
specs on exactly-sized heap buffers against an independent evaluation over the index space, permutations there
and back, the generators, malformed specs.  The oracle checks every GPU result; this checks the checker."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_sanitizers(tmp_path):
    exe = str(tmp_path / "oracle_sanitize")
    cmd = ["gcc", "-std=c11", "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "oracle_sanitize.c"),
           os.path.join(ROOT, "oracle", "ctc_contract_ref.c"), "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr or "ubsan" in r.stderr) and "error:" not in r.stderr:
        pytest.skip("this gcc has no sanitizer runtimes to link")
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="4")
    r = subprocess.run([exe, "1200", "5"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "1200 cases: 0 failures" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])

"""The contraction and permutation planners under AddressSanitizer + UndefinedBehaviorSanitizer (host code only,
gcc): tools/planner_sanitize.cpp draws seeded random specs, runs the offset tables the planners produce in exact
integer arithmetic against a direct evaluation over the index space (whole contractions, slabs of a sharded mode,
permutations), and feeds malformed specs.  A sanitizer report or a wrong table entry fails the run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]


def _build(tmp_path):
    exe = str(tmp_path / "planner_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", *SAN, "-I" + os.path.join(ROOT, "ctc_amd", "csrc"),
           os.path.join(ROOT, "tools", "planner_sanitize.cpp"), os.path.join(ROOT, "ctc_amd", "csrc", "planner.cpp"),
           os.path.join(ROOT, "ctc_amd", "csrc", "permute_plan.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr or "ubsan" in r.stderr) and "error:" not in r.stderr:
        pytest.skip("this gcc has no sanitizer runtimes to link: " + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def test_planners_under_sanitizers(tmp_path):
    exe = _build(tmp_path)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, "1500", "20250101"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert " 0 failures" in r.stdout, r.stdout
    # the harness must notice a wrong table entry (two k offsets exchanged, one repeated)
    r = subprocess.run([exe, "200", "3"], capture_output=True, text=True, env=dict(env, PLANNER_SANITIZE_SELFTEST="1"), timeout=600)
    assert r.returncode == 1 and "tables and direct evaluation differ" in r.stderr, (r.stdout, r.stderr[-500:])

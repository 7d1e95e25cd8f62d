"""The host side of the C ABI (api.cpp, sharded.cpp, fastr_api.cpp, planner.cpp, permute_plan.cpp) compiled with
AddressSanitizer + UndefinedBehaviorSanitizer on the host code, linked with the kernel objects of the ordinary
build and driven by tools/api_sanitize.cpp on this machine (no GPU): seeded random specs through every entry point
that needs no device, short and exact output buffers, invalid arguments; entry points that need a device must come
back with an error code.  Also: sizes that would overflow are refused by the planner."""
import glob
import os
import subprocess

import pytest

import ctc_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_host_side_under_sanitizers(tmp_path):
    if not glob.glob(os.path.join(ROOT, "ctc_amd", "build", "gemm_kernels.o")):
        pytest.skip("no kernel objects of an ordinary build to link with (run __graft_entry__.build() first)")
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "api_sanitize.sh"), "800", "7"], capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "800 cases: 0 failures" in r.stdout, r.stdout[-500:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


def test_sizes_that_would_overflow_are_refused():
    big = 1 << 40
    with pytest.raises(ctc_amd.CtcError, match="too large"):
        ctc_amd.Plan("ab,bc->ac", dict(a=big, b=big, c=big), "f64")
    with pytest.raises(ctc_amd.CtcError, match="too large"):
        ctc_amd.Plan("ab,bc->ac", dict(a=1 << 32, b=4, c=4), "f64")  # 2^32 rows: one table entry per row
    # a stride that reaches past 2^46 elements
    with pytest.raises(ctc_amd.CtcError, match="too large"):
        ctc_amd.Plan("ab,bc->ac", dict(a=1024, b=4, c=4), "f64", strides_a=[1 << 40, 1])


def test_tool_refuses_sizes_it_cannot_hold(tmp_path):
    """ctc_contract with extents no tensor file can have: a message and exit status 2, no abort."""
    tool = os.path.join(ROOT, "ctc_amd", "bin", "ctc_contract")
    big = "a=1099511627776,b=1099511627776"
    r = subprocess.run([tool, "--permute", "ab=ba", "--extents", big, "--a", os.devnull, "--out", str(tmp_path / "x")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "too large" in r.stderr, (r.returncode, r.stderr)
    r = subprocess.run([tool, "--spec", "ab,bc=ac", "--extents", big + ",c=4", "--describe"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "too large" in r.stderr, (r.returncode, r.stderr)

"""The C ABI: the shared library loads without a GPU and exports every symbol the header declares."""
import ctypes
import os
import re

from ctc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="ctc_tc.h", prefix="ctc_tc_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z_0-9]+)\s*\(" % prefix, text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.SYMBOLS + _lib.SHARDED_SYMBOLS)
    assert _declared_symbols("ctc_fastr.h", "ctc_fastr_") == sorted(_lib.FASTR_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols() + _declared_symbols("ctc_fastr.h", "ctc_fastr_"):
        assert hasattr(lib, name), name


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.ctc_tc_version() >= 100
    h = ctypes.c_void_p()
    rc = lib.ctc_tc_plan_create(b"ab,bc", 0, None, None, None, None, None, 0, ctypes.byref(h))
    assert rc < 0
    assert b"spec" in lib.ctc_tc_last_error()


def test_signatures_are_plain_c():
    text = open(os.path.join(ROOT, "include", "ctc_tc.h")).read() + open(os.path.join(ROOT, "include", "ctc_fastr.h")).read()
    assert "torch" not in text and "at::" not in text and "std::" not in text
    assert 'extern "C"' in text


def test_kernels_that_set_m0_from_assembly_have_no_other_m0_user(tmp_path):
    """The LDS-DMA transfers issued from inline assembly (lds_dma.hpp: dma16_split) overwrite M0, which cannot
    be declared as a clobber.  That is safe only while nothing else in those kernels uses M0.  Checked in the
    library's gfx950 code: in every kernel whose transfers come from that assembly, each instruction that
    names M0 is `s_mov_b32 m0, s<N>` directly in front of a `global_load_lds_dwordx4`, each such transfer has
    its own `s_mov_b32 m0` right in front of it, and no instruction that reads M0 implicitly (relative register
    moves, messages, interpolation, GWS) occurs."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        import pytest
        pytest.skip("llvm-objdump is not installed")
    lib = tmp_path / "libctc_tc.so"
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    objs = [p for p in tmp_path.iterdir() if "amdgcn" in p.name]
    assert objs, "no gfx950 code object found in the library"
    # one code object per translation unit of the library
    text = "".join(subprocess.run([objdump, "-d", "--no-show-raw-insn", str(o)], check=True, capture_output=True, text=True).stdout
                   for o in objs if o.stat().st_size > 0)
    kernels, name = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
            kernels[name] = []
        elif name and line.strip():
            kernels[name].append(re.sub(r"\s*//.*$", "", line).strip())
    checked = 0
    for name, insts in kernels.items():
        if "gemm_dma_kernel" not in name and "gemm_bf16_ring_kernel" not in name:
            continue
        checked += 1
        dmas = 0
        for i, inst in enumerate(insts):
            assert not re.match(r"(v_movrel|s_movrel|s_sendmsg|v_interp|ds_gws)", inst), (name, inst)
            if inst.startswith("global_load_lds"):
                dmas += 1
                assert re.match(r"s_mov_b32 m0, s\d+$", insts[i - 1]), (name, insts[i - 1], inst)
            elif re.search(r"\bm0\b", inst):
                assert re.match(r"s_mov_b32 m0, s\d+$", inst) and insts[i + 1].startswith("global_load_lds"), (name, inst, insts[i + 1])
        assert dmas > 0, name
    assert checked >= 16, checked  # fp64 and fp32, four layout pairs, four forms; the bf16 ring kernels

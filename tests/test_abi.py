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

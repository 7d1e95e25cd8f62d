"""ctypes binding of the CPU oracle (oracle/libctc_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and
by bench.py's cpu_baseline leg, never by the ctc_amd package.

PARITY UNPINNED: the reference tree has no contraction path; see
oracle/ctc_contract_ref.h.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libctc_oracle.so")

MAX_MODES = 16
F64, F32, BF16 = 0, 1, 2
CLASS_NAMES = {0: "batch", 1: "contracted", 2: "free_m", 3: "free_n"}
_NP = {F64: np.float64, F32: np.float32, BF16: np.uint16}


class Classification(ctypes.Structure):
    _fields_ = [
        ("n_modes", ctypes.c_int),
        ("names", ctypes.c_char * MAX_MODES),
        ("cls", ctypes.c_int * MAX_MODES),
        ("na", ctypes.c_int), ("nb", ctypes.c_int), ("nc", ctypes.c_int),
        ("a_modes", ctypes.c_int * MAX_MODES),
        ("b_modes", ctypes.c_int * MAX_MODES),
        ("c_modes", ctypes.c_int * MAX_MODES),
        ("n_batch", ctypes.c_int), ("n_m", ctypes.c_int), ("n_n", ctypes.c_int), ("n_k", ctypes.c_int),
        ("batch_modes", ctypes.c_int * MAX_MODES),
        ("m_modes", ctypes.c_int * MAX_MODES),
        ("n_modes_list", ctypes.c_int * MAX_MODES),
        ("k_modes", ctypes.c_int * MAX_MODES),
        ("perm_a", ctypes.c_int * MAX_MODES),
        ("perm_b", ctypes.c_int * MAX_MODES),
        ("perm_c", ctypes.c_int * MAX_MODES),
    ]


def build():
    """Compile the oracle with gcc (idempotent)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libctc_oracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.ctc_ref_splitmix64.restype = ctypes.c_uint64
        L.ctc_ref_splitmix64.argtypes = [ctypes.c_uint64]
        for name in ("ctc_ref_fill_f64", "ctc_ref_fill_f32", "ctc_ref_fill_bf16"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64]
        L.ctc_ref_classify.restype = ctypes.c_int
        L.ctc_ref_classify.argtypes = [ctypes.c_char_p, ctypes.POINTER(Classification),
                                       ctypes.c_char_p, ctypes.c_size_t]
        L.ctc_ref_contract.restype = ctypes.c_int
        L.ctc_ref_contract.argtypes = [
            ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_double, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        L.ctc_ref_permute.restype = ctypes.c_int
        L.ctc_ref_permute.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.ctc_ref_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def splitmix64(x):
    return int(lib().ctc_ref_splitmix64(ctypes.c_uint64(x & 0xFFFFFFFFFFFFFFFF)))


def fill(n, seed, dtype=F64):
    """Synthetic operand: element k = uniform[-1,1) from splitmix64(seed + k)."""
    out = np.empty(int(n), dtype=_NP[dtype])
    fn = {F64: lib().ctc_ref_fill_f64, F32: lib().ctc_ref_fill_f32, BF16: lib().ctc_ref_fill_bf16}[dtype]
    fn(out.ctypes.data, int(n), ctypes.c_uint64(seed))
    return out


def bf16_to_f32(a):
    return (a.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def classify(spec):
    c = Classification()
    err = ctypes.create_string_buffer(256)
    rc = lib().ctc_ref_classify(spec.encode(), ctypes.byref(c), err, 256)
    if rc:
        raise ValueError(err.value.decode())
    names = c.names[:c.n_modes].decode() if isinstance(c.names, bytes) else bytes(c.names)[:c.n_modes].decode()
    return {
        "names": names,
        "classes": {names[i]: CLASS_NAMES[c.cls[i]] for i in range(c.n_modes)},
        "class_codes": [c.cls[i] for i in range(c.n_modes)],
        "a": "".join(names[c.a_modes[i]] for i in range(c.na)),
        "b": "".join(names[c.b_modes[i]] for i in range(c.nb)),
        "c": "".join(names[c.c_modes[i]] for i in range(c.nc)),
        "batch": "".join(names[c.batch_modes[i]] for i in range(c.n_batch)),
        "m": "".join(names[c.m_modes[i]] for i in range(c.n_m)),
        "n": "".join(names[c.n_modes_list[i]] for i in range(c.n_n)),
        "k": "".join(names[c.k_modes[i]] for i in range(c.n_k)),
        "perm_a": [c.perm_a[i] for i in range(c.na)],
        "perm_b": [c.perm_b[i] for i in range(c.nb)],
        "perm_c": [c.perm_c[i] for i in range(c.nc)],
    }


def _i64(seq):
    return np.ascontiguousarray(np.asarray(list(seq), dtype=np.int64))


def contract(spec, extents, A, B, C=None, alpha=1.0, beta=0.0, dtype=F64,
             strides_a=None, strides_b=None, strides_c=None, threads=0):
    """C = alpha * contract(A, B) + beta * C.  extents: dict letter -> extent.
    A, B, C are flat (or any-shape contiguous) numpy arrays of the dtype's storage type."""
    cl = classify(spec)
    names = "".join(extents.keys())
    ext = _i64(extents.values())
    if C is None:
        n = 1
        for ch in cl["c"]:
            n *= int(extents[ch])
        C = np.zeros(n, dtype=_NP[dtype])
    sa = _i64(strides_a) if strides_a is not None else None
    sb = _i64(strides_b) if strides_b is not None else None
    sc = _i64(strides_c) if strides_c is not None else None
    err = ctypes.create_string_buffer(256)
    rc = lib().ctc_ref_contract(
        spec.encode(), len(names), names.encode(), ext.ctypes.data,
        sa.ctypes.data if sa is not None else None,
        sb.ctypes.data if sb is not None else None,
        sc.ctypes.data if sc is not None else None,
        dtype, float(alpha), A.ctypes.data, B.ctypes.data, float(beta), C.ctypes.data,
        int(threads), err, 256)
    if rc:
        raise ValueError(err.value.decode())
    return C


def permute(a, extents_in, perm, threads=0):
    """out mode p = in mode perm[p]; both packed, last mode fastest."""
    a = np.ascontiguousarray(a).reshape(-1)
    out = np.empty_like(a)
    ext = _i64(extents_in)
    pm = np.ascontiguousarray(np.asarray(list(perm), dtype=np.int32))
    rc = lib().ctc_ref_permute(a.ctypes.data, out.ctypes.data, len(ext), ext.ctypes.data,
                               pm.ctypes.data, a.dtype.itemsize, int(threads))
    if rc:
        raise ValueError("ctc_ref_permute failed: %d" % rc)
    return out


def max_threads():
    return int(lib().ctc_ref_max_threads())

"""ctypes binding of libctc_tc.so (the C ABI declared in include/ctc_tc.h).

The library is the product: if it is missing the import fails loudly; there is
no CPU fallback and nothing here touches the oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CTC_TC_LIB points at another build of the same ABI (e.g. a diagnostic build)
LIB_PATH = os.environ.get("CTC_TC_LIB") or os.path.join(_HERE, "lib", "libctc_tc.so")

# every symbol include/ctc_tc.h declares
SYMBOLS = (
    "ctc_tc_version", "ctc_tc_last_error", "ctc_tc_device_count",
    "ctc_tc_plan_create", "ctc_tc_plan_destroy", "ctc_tc_plan_describe", "ctc_tc_plan_get",
    "ctc_tc_plan_shard", "ctc_tc_plan_gemm_plan", "ctc_tc_plan_prepare", "ctc_tc_execute", "ctc_tc_permute",
    "ctc_tc_ttgt_workspace_bytes", "ctc_tc_execute_ttgt", "ctc_tc_fill", "ctc_tc_time_execute", "ctc_tc_time_execute_each",
    "ctc_tc_grouped_contract", "ctc_tc_execute_noalloc", "ctc_tc_noalloc_scratch_bytes", "ctc_tc_set_option", "ctc_tc_get_option",
)
# contracted-mode sharding over RCCL (also declared in include/ctc_tc.h)
SHARDED_SYMBOLS = (
    "ctc_tc_comm_unique_id", "ctc_tc_comm_init_rank", "ctc_tc_comm_init_all", "ctc_tc_comm_destroy", "ctc_tc_comm_abort",
    "ctc_tc_sharded_create", "ctc_tc_sharded_destroy", "ctc_tc_sharded_get", "ctc_tc_sharded_chunk_plan",
    "ctc_tc_sharded_prepare", "ctc_tc_sharded_execute",
)
UNIQUE_ID_BYTES = 128
REDUCE_SCATTER, ALL_REDUCE = 0, 1
SQ_CHUNKS, SQ_C_ELEMS, SQ_OUT_ELEMS, SQ_WORKSPACE_ELEMS = 0, 1, 2, 3
SQ_CHUNK_OFF_A, SQ_CHUNK_OFF_B, SQ_CHUNK_OFF_PARTIAL, SQ_CHUNK_OFF_OUT, SQ_CHUNK_ELEMS = 16, 17, 18, 19, 20
# every symbol include/ctc_fastr.h declares
FASTR_SYMBOLS = ("ctc_fastr_best_offsets", "ctc_fastr_sw_align", "ctc_fastr_sw_workspace_bytes")

F64, F32, BF16 = 0, 1, 2

# ctc_tc_query
Q_NAMES, Q_CLASSES, Q_EXTENTS = 0, 1, 2
Q_CANON_BATCH, Q_CANON_M, Q_CANON_N, Q_CANON_K = 3, 4, 5, 6
Q_PERM_A, Q_PERM_B, Q_PERM_C = 7, 8, 9
Q_SWAPPED, Q_EXEC_BATCH, Q_EXEC_M, Q_EXEC_N, Q_EXEC_K = 10, 11, 12, 13, 14
Q_GEMM_DIMS, Q_TILE = 15, 16
Q_ROW_A, Q_K_A, Q_K_B, Q_COL_B, Q_ROW_C, Q_COL_C, Q_BAT_A, Q_BAT_B, Q_BAT_C = range(17, 26)
Q_FLOPS = 26
Q_SCHEDULE = 27
Q_SPANS = 28


class CtcError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("ctc_tc error %d: %s" % (code, message))
        self.code = code


_lib = None


def load():
    """Load libctc_tc.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libctc_tc.so not found at %s: build it with `make -C ctc_amd/csrc` "
            "(there is no CPU fallback)" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, i64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)
    L.ctc_tc_version.restype = ctypes.c_int
    L.ctc_tc_last_error.restype = ctypes.c_char_p
    L.ctc_tc_device_count.restype = ctypes.c_int
    L.ctc_tc_plan_create.restype = ctypes.c_int
    L.ctc_tc_plan_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, vp, vp, vp, vp,
                                     ctypes.c_int, ctypes.POINTER(vp)]
    L.ctc_tc_plan_destroy.restype = ctypes.c_int
    L.ctc_tc_plan_destroy.argtypes = [vp]
    L.ctc_tc_plan_describe.restype = ctypes.c_int
    L.ctc_tc_plan_describe.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
    L.ctc_tc_plan_get.restype = ctypes.c_int64
    L.ctc_tc_plan_get.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int64]
    L.ctc_tc_plan_gemm_plan.restype = vp
    L.ctc_tc_plan_gemm_plan.argtypes = [vp]
    L.ctc_tc_plan_shard.restype = ctypes.c_int
    L.ctc_tc_plan_shard.argtypes = [vp, ctypes.c_char, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp),
                                    i64p, i64p, i64p]
    L.ctc_tc_plan_prepare.restype = ctypes.c_int
    L.ctc_tc_plan_prepare.argtypes = [vp]
    L.ctc_tc_execute.restype = ctypes.c_int
    L.ctc_tc_execute.argtypes = [vp, ctypes.c_double, vp, vp, ctypes.c_double, vp, vp]
    L.ctc_tc_execute_noalloc.restype = ctypes.c_int
    L.ctc_tc_execute_noalloc.argtypes = [vp, ctypes.c_double, vp, vp, ctypes.c_double, vp, vp]
    L.ctc_tc_noalloc_scratch_bytes.restype = ctypes.c_int64
    L.ctc_tc_get_option.restype = ctypes.c_int64
    L.ctc_tc_get_option.argtypes = [ctypes.c_char_p]
    L.ctc_tc_set_option.restype = ctypes.c_int
    L.ctc_tc_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    L.ctc_tc_permute.restype = ctypes.c_int
    L.ctc_tc_permute.argtypes = [vp, vp, ctypes.c_int, vp, vp, ctypes.c_int, vp]
    L.ctc_tc_ttgt_workspace_bytes.restype = ctypes.c_int64
    L.ctc_tc_ttgt_workspace_bytes.argtypes = [vp]
    L.ctc_tc_execute_ttgt.restype = ctypes.c_int
    L.ctc_tc_execute_ttgt.argtypes = [vp, ctypes.c_double, vp, vp, ctypes.c_double, vp, vp,
                                      ctypes.c_int64, vp]
    L.ctc_tc_fill.restype = ctypes.c_int
    L.ctc_tc_fill.argtypes = [vp, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int, vp]
    L.ctc_tc_grouped_contract.restype = ctypes.c_int
    L.ctc_tc_grouped_contract.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, vp, vp, ctypes.c_double, vp, vp]
    L.ctc_tc_time_execute.restype = ctypes.c_int
    L.ctc_tc_time_execute.argtypes = [vp, ctypes.c_double, vp, vp, ctypes.c_double, vp, vp,
                                      ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    L.ctc_tc_time_execute_each.restype = ctypes.c_int
    L.ctc_tc_time_execute_each.argtypes = [vp, ctypes.c_double, vp, vp, ctypes.c_double, vp, vp,
                                           ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    L.ctc_tc_comm_unique_id.restype = ctypes.c_int
    L.ctc_tc_comm_unique_id.argtypes = [vp]
    L.ctc_tc_comm_init_rank.restype = ctypes.c_int
    L.ctc_tc_comm_init_rank.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, vp]
    L.ctc_tc_comm_init_all.restype = ctypes.c_int
    L.ctc_tc_comm_init_all.argtypes = [ctypes.POINTER(vp), ctypes.c_int, vp]
    L.ctc_tc_comm_destroy.restype = ctypes.c_int
    L.ctc_tc_comm_destroy.argtypes = [vp]
    L.ctc_tc_comm_abort.restype = ctypes.c_int
    L.ctc_tc_comm_abort.argtypes = [vp]
    L.ctc_tc_sharded_create.restype = ctypes.c_int
    L.ctc_tc_sharded_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, vp, ctypes.c_int, ctypes.c_char,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
    L.ctc_tc_sharded_destroy.restype = ctypes.c_int
    L.ctc_tc_sharded_destroy.argtypes = [vp]
    L.ctc_tc_sharded_get.restype = ctypes.c_int64
    L.ctc_tc_sharded_get.argtypes = [vp, ctypes.c_int, ctypes.c_int64]
    L.ctc_tc_sharded_chunk_plan.restype = vp
    L.ctc_tc_sharded_chunk_plan.argtypes = [vp, ctypes.c_int]
    L.ctc_tc_sharded_prepare.restype = ctypes.c_int
    L.ctc_tc_sharded_prepare.argtypes = [vp]
    L.ctc_tc_sharded_execute.restype = ctypes.c_int
    L.ctc_tc_sharded_execute.argtypes = [vp, vp, ctypes.c_double, vp, vp, vp, vp, ctypes.c_int64, vp]
    _lib = L
    return L


def check(rc):
    if rc < 0:
        raise CtcError(rc, load().ctc_tc_last_error().decode())
    return rc

"""Host side of the contraction path: plans, execution, permutation, sharding.

Everything here goes through the C ABI of libctc_tc.so (include/ctc_tc.h).
PyTorch is used only as the owner of device memory and streams; no torch
operator computes any part of a contraction.
"""
import ctypes

from . import _lib
from ._lib import BF16, F32, F64, CtcError, check  # noqa: F401

_DTYPES = {"f64": F64, "f32": F32, "bf16": BF16, F64: F64, F32: F32, BF16: BF16}
DTYPE_SIZE = {F64: 8, F32: 4, BF16: 2}
CLASS_NAMES = {0: "batch", 1: "contracted", 2: "free_m", 3: "free_n"}
LAYOUT_NAMES = {0: "k_vec", 1: "outer_vec", 2: "scalar"}
KERNEL_NAMES = {0: "mfma_gemm", 1: "skinny_same", 2: "skinny_transpose"}


def _dtype_code(dtype):
    try:
        import torch
        if isinstance(dtype, torch.dtype):
            return {torch.float64: F64, torch.float32: F32, torch.bfloat16: BF16}[dtype]
    except ImportError:
        pass
    return _DTYPES[dtype]


def _i64_array(values):
    if values is None:
        return None
    values = [int(v) for v in values]
    return (ctypes.c_int64 * len(values))(*values)


def _ptr(x):
    """Device pointer of a torch tensor, or an int address passed through."""
    if x is None:
        return None
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if not x.is_cuda:
        raise ValueError("operands must live in device memory (got a host tensor)")
    return ctypes.c_void_p(x.data_ptr())


def _room(x):
    """Elements from x's first element to the end of the storage behind it (None for a raw address)."""
    if x is None or isinstance(x, int):
        return None
    return x.untyped_storage().nbytes() // x.element_size() - x.storage_offset()


def _stream(stream):
    if stream is None:
        try:
            import torch
            if torch.cuda.is_available():
                return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        except ImportError:
            pass
        return None
    if isinstance(stream, int):
        return ctypes.c_void_p(stream)
    return ctypes.c_void_p(stream.cuda_stream)


class Plan:
    """A planned contraction C = alpha * A (x) B + beta * C.

    spec     "aeim,ebmj->abij"
    extents  dict letter -> extent
    strides  per operand position, in elements; None = packed, last mode fastest
    """

    def __init__(self, spec, extents, dtype="f64", strides_a=None, strides_b=None, strides_c=None,
                 _handle=None, _owned=True):
        self._lib = _lib.load()
        self._owned = _owned  # False: the handle belongs to another object (a chunk plan of a sharded contraction)
        self.spec = spec
        self.extents = dict(extents)
        self.dtype = _dtype_code(dtype)
        if _handle is not None:
            self._h = _handle
            return
        names = "".join(self.extents.keys()).encode()
        ext = _i64_array(self.extents.values())
        h = ctypes.c_void_p()
        check(self._lib.ctc_tc_plan_create(
            spec.encode(), len(names), names, ext, _i64_array(strides_a), _i64_array(strides_b),
            _i64_array(strides_c), self.dtype, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and self._lib is not None and getattr(self, "_owned", True):
            self._lib.ctc_tc_plan_destroy(h)
        self._h = None

    # ---- planner outputs -------------------------------------------------
    def get(self, query):
        n = check(self._lib.ctc_tc_plan_get(self._h, query, None, 0))
        buf = (ctypes.c_int64 * max(int(n), 1))()
        check(self._lib.ctc_tc_plan_get(self._h, query, buf, n))
        return [int(buf[i]) for i in range(n)]

    def table(self, query):
        """Offset table as a numpy int64 array."""
        import numpy as np
        n = check(self._lib.ctc_tc_plan_get(self._h, query, None, 0))
        out = np.zeros(int(n), dtype=np.int64)
        check(self._lib.ctc_tc_plan_get(self._h, query, out.ctypes.data, n))
        return out

    def describe(self):
        n = check(self._lib.ctc_tc_plan_describe(self._h, None, 0))
        buf = ctypes.create_string_buffer(n + 1)
        check(self._lib.ctc_tc_plan_describe(self._h, buf, n + 1))
        return buf.value.decode()

    @property
    def names(self):
        return "".join(chr(c) for c in self.get(_lib.Q_NAMES))

    @property
    def classes(self):
        return {n: CLASS_NAMES[c] for n, c in zip(self.names, self.get(_lib.Q_CLASSES))}

    def _letters(self, query):
        names = self.names
        return "".join(names[i] for i in self.get(query))

    def summary(self):
        bm, bn, bk, threads, la, lb, va, vb, group, kernel, vc, transposes = self.get(_lib.Q_TILE)
        m, n, k, batch = self.get(_lib.Q_GEMM_DIMS)
        return {
            "names": self.names,
            "class_codes": self.get(_lib.Q_CLASSES),
            "batch": self._letters(_lib.Q_CANON_BATCH), "m": self._letters(_lib.Q_CANON_M),
            "n": self._letters(_lib.Q_CANON_N), "k": self._letters(_lib.Q_CANON_K),
            "perm_a": self.get(_lib.Q_PERM_A), "perm_b": self.get(_lib.Q_PERM_B),
            "perm_c": self.get(_lib.Q_PERM_C),
            "swapped": bool(self.get(_lib.Q_SWAPPED)[0]),
            "exec_batch": self._letters(_lib.Q_EXEC_BATCH), "exec_m": self._letters(_lib.Q_EXEC_M),
            "exec_n": self._letters(_lib.Q_EXEC_N), "exec_k": self._letters(_lib.Q_EXEC_K),
            "M": m, "N": n, "K": k, "batch_count": batch,
            "tile": (bm, bn, bk), "threads": threads,
            "layout_a": LAYOUT_NAMES[la], "layout_b": LAYOUT_NAMES[lb], "vec_a": va, "vec_b": vb, "vec_c": vc,
            "group": group, "kernel": KERNEL_NAMES[kernel],
            "transposes": ("fused", "explicit", "padded_copies")[transposes],
        }

    @property
    def flops(self):
        return self.get(_lib.Q_FLOPS)[0]

    def schedule(self):
        """How the current device will run the tiles (needs a GPU): tiles run whole, tiles of the last,
        partly filled round cut along K, chunks per such tile, workspace bytes."""
        nbm, nbn, whole, tail, split, ws, form, ws_worst, shares, listed = self.get(_lib.Q_SCHEDULE)
        return {"tile_rows": nbm, "tile_cols": nbn, "whole_tiles": whole, "tail_tiles": tail, "split": split,
                "workspace_bytes": ws, "dma_form": form, "workspace_bytes_worst": ws_worst, "stream_k_shares": shares,
                "batch_entries_in_one_tile_list": listed}

    # DMA forms of the tiled GEMM (CTC_TC_Q_SCHEDULE): bytes of k per image row, images per operand and stage, ragged
    _DMA_FORMS = {1: (64, 1, "false"), 2: (32, 2, "false"), 3: (32, 1, "false"), 4: (64, 1, "true")}

    def gemm_kernel_name(self):
        """Name (as a profiler prints it) and K-tile of the tiled-GEMM kernel a launch of this plan with aligned
        operands takes on the current device when neither the persistent nor a skinny kernel claims it."""
        s = self.summary()
        ctype = {F64: "double", F32: "float"}.get(self.dtype)
        form = self.schedule()["dma_form"] if ctype else 0
        if form:
            rb, nsub, ragged = self._DMA_FORMS[form]
            name = "gemm_dma_kernel<%s, %d, %d, %d, %d, %s>" % (ctype, s["layout_a"] != "k_vec", s["layout_b"] != "k_vec",
                                                                  rb, nsub, ragged)
            return name, [128, 128, nsub * rb // DTYPE_SIZE[self.dtype]]
        return ("gemm_kernel<%s>" % ctype if ctype else "gemm_bf16_kernel"), list(s["tile"])

    def gemm_plan(self):
        """The plan whose tiled GEMM execute() launches: this plan, or the sub-plan it owns when it runs through
        explicit transposes or padded copies (summary()["transposes"])."""
        h = self._lib.ctc_tc_plan_gemm_plan(self._h)
        if not h:
            check(-1)
        if h == self._h.value:
            return self
        sub = Plan(self.spec, self.extents, self.dtype, _handle=ctypes.c_void_p(h), _owned=False)
        sub._parent = self  # the handle lives as long as its owner
        return sub

    # ---- sharding ----------------------------------------------------------
    def shard(self, mode, rank, nranks):
        """Sub-plan for slab `rank` of `nranks` along `mode`, plus element offsets into A, B, C."""
        h = ctypes.c_void_p()
        oa, ob, oc = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(self._lib.ctc_tc_plan_shard(self._h, mode.encode(), rank, nranks, ctypes.byref(h),
                                          ctypes.byref(oa), ctypes.byref(ob), ctypes.byref(oc)))
        ext = dict(self.extents)
        ext[mode] //= nranks
        return Plan(self.spec, ext, self.dtype, _handle=h), oa.value, ob.value, oc.value

    # ---- execution -----------------------------------------------------------
    def prepare(self):
        check(self._lib.ctc_tc_plan_prepare(self._h))

    def spans(self):
        """Elements of A, B and C an execution addresses, counted from the pointers it is given."""
        if getattr(self, "_spans", None) is None:
            self._spans = tuple(self.get(_lib.Q_SPANS))
        return self._spans

    def _check_operands(self, A, B, C):
        """The kernels trust their offset tables: a buffer shorter than the plan addresses, or of another
        element size, would be read or written past its end on the device.  Tensors are checked here, before
        anything is launched; raw addresses (ints) cannot be."""
        for name, x, need in zip("ABC", (A, B, C), self.spans()):
            if x is None:
                raise ValueError("operand %s is None" % name)
            if isinstance(x, int):
                continue
            if x.element_size() != DTYPE_SIZE[self.dtype]:
                raise ValueError("operand %s has %d-byte elements, the plan was made for %d-byte ones"
                                 % (name, x.element_size(), DTYPE_SIZE[self.dtype]))
            if _room(x) < need:
                raise ValueError("operand %s holds %d elements from its first one on, the plan addresses %d"
                                 % (name, _room(x), need))

    def execute(self, A, B, C, alpha=1.0, beta=0.0, stream=None):
        self._check_operands(A, B, C)
        check(self._lib.ctc_tc_execute(self._h, float(alpha), _ptr(A), _ptr(B), float(beta), _ptr(C),
                                       _stream(stream)))
        return C

    def execute_noalloc(self, A, B, C, alpha=1.0, beta=0.0, stream=None):
        """execute() through the library's fixed per-GPU scratch buffer: no device allocation.  Needs
        prepare() first; calls that overlap on one GPU race (they share the buffer)."""
        self._check_operands(A, B, C)
        check(self._lib.ctc_tc_execute_noalloc(self._h, float(alpha), _ptr(A), _ptr(B), float(beta), _ptr(C),
                                               _stream(stream)))
        return C

    def ttgt_workspace_bytes(self):
        return int(check(self._lib.ctc_tc_ttgt_workspace_bytes(self._h)))

    def execute_ttgt(self, A, B, C, workspace, alpha=1.0, beta=0.0, stream=None):
        self._check_operands(A, B, C)
        nbytes = min(workspace.numel(), _room(workspace)) * workspace.element_size()
        check(self._lib.ctc_tc_execute_ttgt(self._h, float(alpha), _ptr(A), _ptr(B), float(beta), _ptr(C),
                                            _ptr(workspace), nbytes, _stream(stream)))
        return C

    def time(self, A, B, C, warmup=2, iters=10, alpha=1.0, beta=0.0, stream=None):
        """Average device milliseconds per execution, from HIP events on the launch stream."""
        self._check_operands(A, B, C)
        ms = ctypes.c_double()
        check(self._lib.ctc_tc_time_execute(self._h, float(alpha), _ptr(A), _ptr(B), float(beta), _ptr(C),
                                            _stream(stream), warmup, iters, ctypes.byref(ms)))
        return ms.value

    def time_each(self, A, B, C, iters=10, alpha=1.0, beta=0.0, stream=None):
        """Device milliseconds of each of `iters` back-to-back executions (an event between every two)."""
        self._check_operands(A, B, C)
        ms = (ctypes.c_double * iters)()
        check(self._lib.ctc_tc_time_execute_each(self._h, float(alpha), _ptr(A), _ptr(B), float(beta), _ptr(C),
                                                 _stream(stream), iters, ms))
        return [float(x) for x in ms]


def contract(spec, A, B, out=None, alpha=1.0, beta=0.0, stream=None):
    """Contract two device tensors (torch, cuda).  Extents and strides come from the tensors."""
    import torch
    ins, outm = spec.replace(" ", "").replace("->", "=").split("=")
    sa, sb = ins.split(",")
    if len(sa) != A.dim() or len(sb) != B.dim():
        raise ValueError("spec rank does not match the operands")
    extents = {}
    for letters, t in ((sa, A), (sb, B)):
        for ch, e in zip(letters, t.shape):
            if extents.setdefault(ch, int(e)) != int(e):
                raise ValueError("extent mismatch on mode %r" % ch)
    if out is None:
        out = torch.empty([extents[ch] for ch in outm], dtype=A.dtype, device=A.device)
    plan = Plan(spec, extents, A.dtype, A.stride(), B.stride(), out.stride())
    plan.execute(A, B, out, alpha, beta, stream)
    return out


def permute(x, perm, out=None, stream=None):
    """out mode p = x mode perm[p]; x packed (contiguous); result packed."""
    import torch
    if not x.is_contiguous():
        raise ValueError("permute expects a packed input")
    L = _lib.load()
    perm = [int(p) for p in perm]
    if out is None:
        out = torch.empty([x.shape[p] for p in perm], dtype=x.dtype, device=x.device)
    elif out.element_size() != x.element_size() or _room(out) < x.numel() or not out.is_contiguous():
        raise ValueError("permute: the result must be packed, of the input's element size and hold %d elements" % x.numel())
    ext = _i64_array(x.shape)
    pm = (ctypes.c_int * len(perm))(*perm)
    check(L.ctc_tc_permute(_ptr(x), _ptr(out), x.dim(), ext, pm, x.element_size(), _stream(stream)))
    return out


def fill(x, seed, first=0, stream=None):
    """Fill a packed device tensor with the synthetic stream uniform[-1,1) of splitmix64(seed + first + k)."""
    L = _lib.load()
    if not x.is_contiguous():
        raise ValueError("fill expects a packed tensor")
    check(L.ctc_tc_fill(_ptr(x), x.numel(), ctypes.c_uint64(seed), int(first), _dtype_code(x.dtype),
                        _stream(stream)))
    return x


SMALL_GEMM_DTYPE = [("m", "<i4"), ("n", "<i4"), ("k", "<i4"), ("lda", "<i4"), ("ldb", "<i4"), ("ldc", "<i4"),
                    ("off_a", "<i8"), ("off_b", "<i8"), ("off_c", "<i8")]


def pack_small_problems(dims):
    """Descriptors (numpy structured array, layout of ctc_tc_small_gemm) for packed row-major problems
    laid out back to back; dims is a sequence of (m, n, k).  Returns (descriptors, |A|, |B|, |C|)."""
    import numpy as np
    d = np.zeros(len(dims), dtype=SMALL_GEMM_DTYPE)
    oa = ob = oc = 0
    for i, (m, n, k) in enumerate(dims):
        d[i] = (m, n, k, k, n, n, oa, ob, oc)
        oa += m * k
        ob += k * n
        oc += m * n
    return d, oa, ob, oc


def grouped_contract(problems, A, B, C, alpha=1.0, beta=0.0, stream=None):
    """All small contractions ab = ak,kb described by `problems` (device tensor holding
    ctc_tc_small_gemm descriptors, e.g. torch.from_numpy(desc.view(numpy.uint8)).cuda()) in one launch."""
    L = _lib.load()
    count = problems.numel() * problems.element_size() // 48
    check(L.ctc_tc_grouped_contract(_ptr(problems), count, _dtype_code(A.dtype), float(alpha), _ptr(A), _ptr(B),
                                    float(beta), _ptr(C), _stream(stream)))
    return C


def set_option(name, value):
    """Process-wide kernel-selection knob (ctc_tc_set_option): "stream_mode" 0 auto / 1 never / 2 whenever
    possible, "stream_max_k"."""
    check(_lib.load().ctc_tc_set_option(name.encode(), int(value)))


def get_option(name):
    """Value of a knob, or of a launch counter ("launches_classic", "launches_dma_gemm", "launches_stream",
    "launches_stream_f64_dma")."""
    return int(check(_lib.load().ctc_tc_get_option(name.encode())))


def noalloc_scratch_bytes():
    """Size of the fixed per-GPU scratch buffer Plan.execute_noalloc works in."""
    return int(_lib.load().ctc_tc_noalloc_scratch_bytes())


def device_count():
    return int(_lib.load().ctc_tc_device_count())

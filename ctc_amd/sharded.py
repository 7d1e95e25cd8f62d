"""Multi-GPU execution of one contraction: one process per GPU.

Free or batch mode:  every rank computes a disjoint slab of C from its slab of the operand
                     that holds the mode; the other operand is replicated.  No collective.
Contracted mode:     every rank holds 1/p of A and of B along the mode and produces a partial C;
                     the partials are summed inside libctc_tc.so (ctc_tc_sharded_*, RCCL called
                     from C): reduce-scatter (each rank keeps the flat range r of C, the natural
                     form over xGMI's point-to-point links) or all-reduce (C replicated), cut into
                     chunks so that the sum of chunk i overlaps the GEMM of chunk i + 1.

Operands are rank-local packed slabs (extent/p along the sharded mode).  torch.distributed is used
for ONE thing here: handing rank 0's RCCL unique id to the other ranks.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib
from .host import CLASS_NAMES, Plan, _dtype_code, _i64_array, _ptr, _room, _stream, check  # noqa: F401


def _numel(letters, extents):
    n = 1
    for ch in letters:
        n *= int(extents[ch])
    return n


class Communicator:
    """An RCCL communicator owned by libctc_tc.so (ncclComm_t behind a void*), one rank per process.
    The unique id travels through torch.distributed (any backend; gloo will do)."""

    def __init__(self, rank, world, group=None):
        L = _lib.load()
        self._lib, self.rank, self.world = L, rank, world
        buf = (ctypes.c_ubyte * _lib.UNIQUE_ID_BYTES)()
        if rank == 0:
            check(L.ctc_tc_comm_unique_id(buf))
        if world > 1:
            box = [bytes(buf)]
            dist.broadcast_object_list(box, src=0, group=group)
            buf = (ctypes.c_ubyte * _lib.UNIQUE_ID_BYTES).from_buffer_copy(box[0])
        h = ctypes.c_void_p()
        check(L.ctc_tc_comm_init_rank(ctypes.byref(h), world, rank, buf))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and self._lib is not None:
            self._lib.ctc_tc_comm_destroy(h)
            self._h = None


class ShardedContraction:
    """Plan of rank `rank` of `world` for `spec` sharded along `mode`.

    nchunks: pipeline depth of the contracted-mode sum (1: one collective after one GEMM)."""

    def __init__(self, spec, extents, dtype, mode, rank, world, reduce="reduce_scatter", nchunks=4):
        self.spec, self.mode, self.rank, self.world = spec, mode, rank, world
        self._lib = _lib.load()
        full = Plan(spec, extents, dtype)
        self.dtype = full.dtype
        self.kind = full.classes[mode]  # "free_m" / "free_n" / "batch" / "contracted"
        if extents[mode] % world:
            raise ValueError("extent of mode %r (%d) is not a multiple of the world size %d"
                             % (mode, extents[mode], world))
        self.local_extents = dict(extents)
        self.local_extents[mode] //= world
        self.first = rank * self.local_extents[mode]      # first index of this rank's slab
        self.plan = Plan(spec, self.local_extents, dtype)  # packed rank-local operands
        ins, self.c_letters = spec.replace(" ", "").replace("->", "=").split("=")
        self.a_letters, self.b_letters = ins.split(",")
        self.extents = dict(extents)
        self.reduce = reduce
        self._h = None
        if self.kind == "contracted":
            if reduce not in ("reduce_scatter", "all_reduce"):
                raise ValueError("reduce must be 'reduce_scatter' or 'all_reduce'")
            names = "".join(self.extents.keys()).encode()
            h = ctypes.c_void_p()
            check(self._lib.ctc_tc_sharded_create(
                spec.encode(), len(names), names, _i64_array(self.extents.values()), self.dtype, mode.encode(), rank, world,
                int(nchunks), _lib.REDUCE_SCATTER if reduce == "reduce_scatter" else _lib.ALL_REDUCE, ctypes.byref(h)))
            self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and self._lib is not None:
            self._lib.ctc_tc_sharded_destroy(h)
            self._h = None

    # sizes of what a rank holds
    def local_numel(self, which):
        letters = {"a": self.a_letters, "b": self.b_letters, "c": self.c_letters}[which]
        ext = self.local_extents if (self.mode in letters) else self.extents
        return _numel(letters, ext)

    def full_c_numel(self):
        return _numel(self.c_letters, self.extents)

    def out_numel(self):
        """Elements of the buffer execute() fills on this rank."""
        if self.kind != "contracted":
            return self.local_numel("c")
        return int(self._get(_lib.SQ_OUT_ELEMS))

    def workspace_numel(self):
        return int(self._get(_lib.SQ_WORKSPACE_ELEMS)) if self.kind == "contracted" else 0

    def _get(self, query, chunk=0):
        return check(self._lib.ctc_tc_sharded_get(self._h, query, chunk))

    @property
    def nchunks(self):
        return int(self._get(_lib.SQ_CHUNKS)) if self.kind == "contracted" else 1

    def chunk(self, c):
        """(plan, off_a, off_b, off_partial, off_out, elems) of chunk c of the contracted-mode pipeline."""
        h = self._lib.ctc_tc_sharded_chunk_plan(self._h, c)
        if not h:
            check(-1)
        plan = Plan(self.spec, self.local_extents, self.dtype, _handle=ctypes.c_void_p(h), _owned=False)
        return (plan, int(self._get(_lib.SQ_CHUNK_OFF_A, c)), int(self._get(_lib.SQ_CHUNK_OFF_B, c)),
                int(self._get(_lib.SQ_CHUNK_OFF_PARTIAL, c)), int(self._get(_lib.SQ_CHUNK_OFF_OUT, c)),
                int(self._get(_lib.SQ_CHUNK_ELEMS, c)))

    def prepare(self):
        if self.kind == "contracted":
            check(self._lib.ctc_tc_sharded_prepare(self._h))
        else:
            self.plan.prepare()

    def slab_of(self, full_tensor, letters):
        """This rank's packed slab of a full packed tensor (host-side helper for tests and set-up)."""
        if self.mode not in letters:
            return full_tensor
        shape = [self.extents[ch] for ch in letters]
        axis = letters.index(self.mode)
        t = full_tensor.reshape(shape).narrow(axis, self.first, self.local_extents[self.mode])
        return t.contiguous().reshape(-1)

    def execute(self, A, B, C, alpha=1.0, beta=0.0, comm=None, group=None, compute=None, scratch=None, stream=None):
        """Run this rank's part.

        free/batch mode: C is the rank's slab of the result (local_numel('c') elements).
        contracted mode: C receives this rank's flat 1/p of the result (reduce_scatter) or the whole
        result (all_reduce), out_numel() elements; `comm` is a Communicator (may be None when world
        == 1); `scratch` may carry the workspace_numel() elements of the partial result.
        `compute(plan, A, B, C, alpha, beta)` replaces the kernel launch AND moves the collectives to
        torch.distributed on `group` (CPU tests of the sharding logic only: same chunks, same
        offsets, gloo instead of RCCL)."""
        if self.kind != "contracted":
            run = compute or (lambda plan, a, b, c, al, be: plan.execute(a, b, c, al, be, stream))
            run(self.plan, A, B, C, alpha, beta)
            return C
        if beta != 0.0:
            raise ValueError("beta must be 0 when sharding a contracted mode (C would be added p times)")
        scatter = self.reduce == "reduce_scatter"
        n_ws = self.workspace_numel()
        partial = None
        if scatter:
            partial = scratch if scratch is not None else torch.empty(n_ws, dtype=C.dtype, device=C.device)
            if partial.numel() < n_ws:
                raise ValueError("scratch holds %d elements, %d needed" % (partial.numel(), n_ws))
        if compute is None:
            # the C layer takes raw pointers: sizes are checked here, before anything is launched
            for name, x, need in (("A", A, self.local_numel("a")), ("B", B, self.local_numel("b")), ("C", C, self.out_numel())):
                if x.element_size() != C.element_size() or _room(x) < need:
                    raise ValueError("operand %s holds %d elements of %d bytes, this rank's part needs %d elements of the "
                                     "result's element size" % (name, _room(x), x.element_size(), need))
            ws_bytes = n_ws * C.element_size() if scatter else 0
            check(self._lib.ctc_tc_sharded_execute(
                self._h, comm._h if comm is not None else None, float(alpha), _ptr(A), _ptr(B), _ptr(C),
                _ptr(partial) if scatter else None, ws_bytes, _stream(stream)))
            return C
        # emulation: the chunks of the C layer one after the other, collectives through torch.distributed
        for c in range(self.nchunks):
            plan, oa, ob, op, oo, elems = self.chunk(c)
            target = partial if scatter else C
            piece = target[op:op + elems]
            compute(plan, A[oa:], B[ob:], piece, alpha, 0.0)
            if self.world > 1:
                dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=group)
            if scatter:
                per = elems // self.world
                C[oo:oo + per].copy_(piece[self.rank * per:(self.rank + 1) * per])
        return C

"""Multi-GPU execution of one contraction: one process per GPU, torch.distributed (RCCL).

Free or batch mode:  every rank computes a disjoint slab of C from its slab of the operand
                     that holds the mode; the other operand is replicated.  No collective.
Contracted mode:     every rank holds 1/p of A and of B along the mode and produces a
                     full-size partial C; the partials are summed with ONE collective -
                     reduce-scatter (each rank keeps |C|/p, the natural form over xGMI's
                     point-to-point links) or all-reduce (C replicated).

Operands are rank-local packed slabs (extent/p along the sharded mode), which is what a
slab plan of ctc_tc_plan_shard computes on when the mode is the operand's slowest one,
and what Plan(spec, local_extents) computes on in general.
"""
import torch
import torch.distributed as dist

from .host import CLASS_NAMES, Plan  # noqa: F401


def _numel(letters, extents):
    n = 1
    for ch in letters:
        n *= int(extents[ch])
    return n


class ShardedContraction:
    """Plan of rank `rank` of `world` for `spec` sharded along `mode`."""

    def __init__(self, spec, extents, dtype, mode, rank, world):
        self.spec, self.mode, self.rank, self.world = spec, mode, rank, world
        full = Plan(spec, extents, dtype)
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

    # sizes of what a rank holds
    def local_numel(self, which):
        letters = {"a": self.a_letters, "b": self.b_letters, "c": self.c_letters}[which]
        ext = self.local_extents if (self.mode in letters) else self.extents
        return _numel(letters, ext)

    def full_c_numel(self):
        return _numel(self.c_letters, self.extents)

    def slab_of(self, full_tensor, letters):
        """This rank's packed slab of a full packed tensor (host-side helper for tests and set-up)."""
        if self.mode not in letters:
            return full_tensor
        shape = [self.extents[ch] for ch in letters]
        axis = letters.index(self.mode)
        t = full_tensor.reshape(shape).narrow(axis, self.first, self.local_extents[self.mode])
        return t.contiguous().reshape(-1)

    def execute(self, A, B, C, alpha=1.0, beta=0.0, group=None, reduce="reduce_scatter", compute=None,
                scratch=None):
        """Run this rank's part.

        free/batch mode: C is the rank's slab of the result (local_numel('c') elements).
        contracted mode: C receives this rank's 1/p chunk of the flat result (reduce_scatter,
        full_c_numel()/world elements) or the whole result (all_reduce).  `scratch` may carry a
        full-size buffer for the partial result.  `compute(plan, A, B, C, alpha, beta)` replaces
        the kernel launch (used by CPU tests of the sharding logic only)."""
        run = compute or (lambda plan, a, b, c, al, be: plan.execute(a, b, c, al, be))
        if self.kind != "contracted":
            run(self.plan, A, B, C, alpha, beta)
            return C
        if beta != 0.0:
            raise ValueError("beta must be 0 when sharding a contracted mode (C would be added p times)")
        n = self.full_c_numel()
        if reduce == "all_reduce":
            run(self.plan, A, B, C, alpha, 0.0)
            if self.world > 1:
                if C.is_cuda and dist.get_backend(group) == "gloo":
                    host = C.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                    C.copy_(host)
                else:
                    dist.all_reduce(C, op=dist.ReduceOp.SUM, group=group)
            return C
        if n % self.world:
            raise ValueError("reduce_scatter needs |C| divisible by the world size")
        partial = scratch if scratch is not None else torch.empty(n, dtype=C.dtype, device=C.device)
        run(self.plan, A, B, partial, alpha, 0.0)
        if self.world == 1:
            C.copy_(partial)
        elif dist.get_backend(group) == "gloo":
            # gloo has no reduce_scatter: same result through all_reduce + slice, staged through
            # host memory when the partial lives on a GPU (CPU tests and one-GPU rehearsals only)
            host = partial if not partial.is_cuda else partial.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            chunk = n // self.world
            C.copy_(host[self.rank * chunk:(self.rank + 1) * chunk])
        else:
            dist.reduce_scatter_tensor(C, partial, op=dist.ReduceOp.SUM, group=group)
        return C

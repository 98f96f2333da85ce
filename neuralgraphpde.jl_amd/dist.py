"""Data parallelism over independent graph trajectories.

The reference has no multi-process code; a batched GNNGraph is block-diagonal
(/root/reference/test/runtests.jl:89-102, src/layers.jl:359-361), messages never cross graphs, so whole
trajectories shard across GPUs with replicated parameters and ONE all-reduce(sum) of the flat
parameter-gradient vector per backward pass (33 KB for 2 x GCNConv(64=>64): latency bound; RCCL picks
its low-latency protocol for this size on the fully connected xGMI mesh).  One process per GPU;
backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous range of whole trajectories owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def flatten_grads(tree):
    """Flat fp32 vector of all gradient leaves of a nested dict (ComponentArray order: insertion order)."""
    leaves = []

    def walk(t):
        for v in t.values():
            if isinstance(v, dict):
                walk(v)
            else:
                leaves.append(v)
    walk(tree)
    return torch.cat([g.reshape(-1) for g in leaves]), leaves


def allreduce_gradients(tree, group=None, average=False):
    """In-place all-reduce(sum) of every gradient leaf of `tree` (a nested dict of tensors) as ONE
    collective on the flat vector; with average=True divides by the world size afterwards."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tree
    flat, leaves = flatten_grads(tree)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for g in leaves:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return tree


class OverlappedGradReduce:
    """All-reduce of the flat gradient in buckets, each launched on a side stream as soon as autograd has finished the last
    leaf of the bucket -- so that the collective of the gradients that are final EARLY in the backward pass (MPPDEConv: the
    node update psi, whose pullback runs first) overlaps the rest of the pullback (the message MLP phi).  GNOConv's ~4 MB
    gradient is bandwidth- rather than latency-bound on the xGMI ring, which is where the overlap pays; the 33 KB of the GCN
    solver is one latency-bound collective either way.

        flat, ps = optim.flatten_parameters(ps)
        red = OverlappedGradReduce(flat, [("psi.", ...), ("phi.", ...)])     # name prefixes, in the order they become final
        loss.backward(); red.finish()                                         # the current stream now sees the reduced sums
        optim.update(st_opt, flat, reduced=True)

    Without an initialised process group (or world size 1) it does nothing."""

    def __init__(self, flat, ps_views, bucket_prefixes, group=None):
        self.flat, self.group = flat, group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.cuda = flat.grad.is_cuda
        self.side = torch.cuda.Stream() if (self.active and self.cuda) else None
        named = {}

        def walk(t, prefix):
            for k, v in t.items():
                if isinstance(v, dict):
                    walk(v, prefix + k + ".")
                else:
                    named[prefix + k] = v
        walk(ps_views, "")
        self.buckets = []
        taken = set()
        for prefixes in bucket_prefixes:
            prefixes = (prefixes,) if isinstance(prefixes, str) else tuple(prefixes)
            rows = [(name, off, n) for (name, off, shape) in flat.table for n in [int(torch.Size(shape).numel())]
                    if name.startswith(prefixes) and name not in taken]
            if not rows:
                continue
            taken.update(r[0] for r in rows)
            lo, hi = min(r[1] for r in rows), max(r[1] + r[2] for r in rows)
            if sum(r[2] for r in rows) != hi - lo:
                raise ValueError(f"bucket {prefixes} is not a contiguous range of the flat vector")
            self.buckets.append({"names": [r[0] for r in rows], "lo": lo, "hi": hi, "seen": 0, "work": None})
        rest = [(name, off, int(torch.Size(shape).numel())) for (name, off, shape) in flat.table if name not in taken]
        for name, off, n in rest:                      # leaves nobody named: one bucket each, reduced when they arrive
            self.buckets.append({"names": [name], "lo": off, "hi": off + n, "seen": 0, "work": None})
        self._hooks = []
        if self.active:
            for b in self.buckets:
                for name in b["names"]:
                    self._hooks.append(named[name].register_post_accumulate_grad_hook(lambda _p, b=b: self._arrived(b)))

    def _launch(self, b):
        seg = self.flat.grad[b["lo"]:b["hi"]]
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())       # the gradient kernels of this bucket are enqueued
            with torch.cuda.stream(self.side):
                b["work"] = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b["work"] = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _arrived(self, b):
        b["seen"] += 1
        if b["seen"] == len(b["names"]):
            self._launch(b)

    def finish(self):
        """reduce whatever has not been launched (leaves without a gradient this step), wait, reset"""
        if not self.active:
            return
        for b in self.buckets:
            if b["work"] is None:
                self._launch(b)
        for b in self.buckets:
            b["work"].wait()
            b["work"], b["seen"] = None, 0
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


class NativeComm:
    """The library's own communicator (include/ngpde.h: ngpde_comm_*, RCCL over xGMI behind the C ABI) -- what a Julia host binds for
    the data-parallel step; the Python host's default stays torch.distributed.  One rank per GPU, on the current device.

    unique_id: bytes made by rank 0 (NativeComm.unique_id()) and shipped to the others by the host; with torch.distributed already
    initialised, NativeComm.from_torch() does that with a broadcast."""

    def __init__(self, unique_id, rank, world):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        self.lib = _lib.load()
        self.ptr = None
        out = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _lib.check(self.lib.ngpde_comm_create(buf, int(rank), int(world), C.byref(out)))
        self.ptr, self.rank, self.world = out, int(rank), int(world)

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().ngpde_comm_unique_id(buf, 128))
        return bytes(buf.raw)

    @classmethod
    def from_torch(cls, group=None):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(box[0], rank, world)

    def all_reduce(self, flat):
        """in-place sum over the ranks of a contiguous fp32 device vector, on the current stream"""
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        self._lib.check(self.lib.ngpde_grad_allreduce(self.ptr, self._lib.ptr(flat), flat.numel(), self._lib.current_stream()))
        return flat

    def all_reduce_adam(self, x, grad, m, v, eta, beta1, beta2, eps, step):
        """all-reduce(sum) of grad, then the fused Adam step with 1 / world folded in: two launches on the current stream"""
        p = self._lib.ptr
        self._lib.check(self.lib.ngpde_grad_allreduce_adam(self.ptr, x.numel(), p(x), p(grad), p(m), p(v), eta, beta1, beta2, eps, int(step),
                                                           self._lib.current_stream()))

    def rccl_count_and_rank(self):
        """(ncclCommCount, ncclCommUserRank) of the communicator: what RCCL itself says about who met"""
        C = self._C
        cnt, ur = C.c_int32(), C.c_int32()
        self._lib.check(self.lib.ngpde_comm_rccl_info(self.ptr, C.byref(cnt), C.byref(ur)))
        return int(cnt.value), int(ur.value)

    def close(self):
        if self.ptr is not None:
            self.lib.ngpde_comm_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""The device-resident solver plans behind NeuralODE (node.py): handles of the library's plan objects and the autograd functions that
call their forward / adjoint entries.  `_Plan` = ngpde_node_gcn2_* (the two-GCNConv chain of graph_node.md:78, BASELINE's headline);
`_OdePlan` = ngpde_ode_* (ONE create call: the library is given a description of the right-hand side and picks the plan -- a GAT-style
layer, VMHConv(phi, gamma) or the GCN chain; csrc/api_ode.hip).  A plan holds the tape of ONE solve: `claim` / `busy` hand it out."""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import _lib

class _Token:
    __slots__ = ("__weakref__",)


class _RowsIndexFn(torch.autograd.Function):
    """rows of a [n] / [T][n] float32 array by an index list through ngpde_rows_index (gather, or scatter into zeros): a padded batch's
    state on its way into and out of the device-resident VMH plan -- a library launch, its pullback the opposite one"""

    @staticmethod
    def forward(ctx, x, index, n_rows, scatter):
        x = x.contiguous()
        outer = 1 if x.dim() == 1 else x.shape[0]
        n_idx = int(index.numel())
        out = torch.empty((n_rows if scatter else n_idx,) if x.dim() == 1 else (outer, n_rows if scatter else n_idx), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().ngpde_rows_index(outer, n_rows, n_idx, 1, _lib.ptr(index), _lib.ptr(x), _lib.ptr(out), int(scatter), _lib.current_stream()))
        ctx.meta = (index, n_rows, scatter)
        return out

    @staticmethod
    def backward(ctx, g):
        index, n_rows, scatter = ctx.meta
        return _RowsIndexFn.apply(g, index, n_rows, not scatter), None, None, None


def _rows_index(x, index, n_rows, scatter):
    return _RowsIndexFn.apply(x, index, int(n_rows), bool(scatter))


def _check_plan_shapes(what, u, n_rows, d, weights, biases):
    """DimensionMismatch (the reference's error for these: check_num_nodes / the matrix product) unless u is [n_rows][d], every
    weight has its shape and every bias its length -- the device-resident plans' C entries take pointers only"""
    if tuple(u.shape) != (n_rows, d):
        raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, f"DimensionMismatch: {what}: the state is ({u.shape[1]} x {u.shape[0]}), "
                                     f"the graph and layers need ({d} x {n_rows})")
    for name, w, shape in weights:
        if tuple(w.shape) != tuple(shape):
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, f"DimensionMismatch: {what}: {name} is {tuple(w.shape)[::-1]}, expected {shape[::-1]}")
    for name, b, n in biases:
        if b is not None and b.numel() != n:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, f"DimensionMismatch: {what}: {name} has {b.numel()} entries, expected {n}")


class _Plan:
    def __init__(self, handle, d, act, tableau, n_steps, dt, with_backward, members=1):
        """members > 1: `handle` is ONE member of a block-diagonal batch of `members` identical structures; the plan takes
        [members * N][d] arrays and solves the members one after the other (ngpde_node_gcn2_create_batch; raises NgpdeError
        with code ERR_UNSUPPORTED when the persistent plan does not cover the case)"""
        self.lib = _lib.load()
        _lib.flush_destroy()            # plans whose finaliser ran inside a HIP-graph capture
        self.handle = handle            # keeps the graph handle alive
        self.ptr = None
        self.members = int(members)
        self.n_nodes = int(handle._n_nodes)     # rows of ONE member
        out = C.c_void_p()
        _lib.check(self.lib.ngpde_node_gcn2_create_batch(handle.ptr, self.members, d, act, _lib.TABLEAU[tableau], n_steps, dt,
                                                         int(with_backward), C.byref(out)))
        self.ptr = out

    def tape_bytes(self):
        return int(self.lib.ngpde_node_tape_bytes(self.ptr))

    def launch_count(self):
        f, b = C.c_int32(), C.c_int32()
        _lib.check(self.lib.ngpde_node_launch_count(self.ptr, C.byref(f), C.byref(b)))
        return f.value, b.value

    def flags(self):
        """the internal forms the plan chose (ngpde_node_flags; names in _FLAG_NAMES)"""
        f = C.c_int32()
        _lib.check(self.lib.ngpde_node_flags(self.ptr, C.byref(f)))
        return {name for bit, name in _FLAG_NAMES if f.value & bit}

    def fault(self):
        """True when a persistent launch of this plan gave up waiting (its outputs are NaN).  Synchronises."""
        f = C.c_int32()
        _lib.check(self.lib.ngpde_node_fault(self.ptr, _lib.current_stream(), C.byref(f)))
        return bool(f.value)

    def claim(self):
        """token held by the autograd node of the solve that now owns the tape; the plan is busy while that node is alive
        and its backward has not run"""
        self._token = _Token()
        self._token_ref = weakref.ref(self._token)
        token, self._token = self._token, None
        return token

    def busy(self):
        ref = getattr(self, "_token_ref", None)
        return ref is not None and ref() is not None and self.generation()[1]

    def generation(self):
        """(generation of the last forward, is its backward still outstanding?) -- ngpde_node_generation"""
        gen, pend = C.c_uint64(), C.c_int32()
        _lib.check(self.lib.ngpde_node_generation(self.ptr, C.byref(gen), C.byref(pend)))
        return gen.value, bool(pend.value)

    def __del__(self):
        try:
            if self.ptr:
                _lib.destroy_later("ngpde_node_destroy", self.ptr)      # (not inside a HIP-graph capture: see _lib.destroy_later)
                self.ptr = None
        except Exception:
            pass


class _NodeGCN2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, w1t, b1, w2t, b2, plan):
        lib = _lib.load()
        u, w1t, w2t = u.contiguous(), w1t.contiguous(), w2t.contiguous()
        uT = torch.empty_like(u)
        _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, _lib.ptr(u), _lib.ptr(w1t), _lib.ptr(b1), _lib.ptr(w2t),
                                               _lib.ptr(b2), _lib.ptr(uT), _lib.current_stream()))
        ctx.plan = plan
        ctx.token = plan.claim()                   # dies with this autograd node: a solve nobody can differentiate any more
        ctx.generation = plan.generation()[0]      # the plan's single tape now belongs to THIS solve
        ctx.shapes = (u.shape, w1t.shape, None if b1 is None else b1.shape, None if b2 is None else b2.shape)
        ctx.dev = u.device
        return uT

    @staticmethod
    def backward(ctx, duT):
        lib = _lib.load()
        us, ws, b1s, b2s = ctx.shapes
        mk = lambda s: None if s is None else torch.empty(s, dtype=torch.float32, device=ctx.dev)
        du0, dw1, dw2, db1, db2 = mk(us), mk(ws), mk(ws), mk(b1s), mk(b2s)
        duT = duT.contiguous()
        # NGPDE_ERR_STATE if another forward has overwritten this solve's tape (cannot happen through NeuralODE.__call__,
        # which takes a free plan for every outstanding solve; a second backward of the same solve is fine: the tape is kept)
        _lib.check(lib.ngpde_node_expect_generation(ctx.plan.ptr, ctx.generation))
        _lib.check(lib.ngpde_node_gcn2_backward(ctx.plan.ptr, _lib.ptr(duT), _lib.ptr(du0), _lib.ptr(dw1), _lib.ptr(db1),
                                                _lib.ptr(dw2), _lib.ptr(db2), _lib.current_stream()))
        return du0, dw1, db1, dw2, db2, None


_FLAG_NAMES = ((1, "prescaled"), (2, "sign_masks"), (4, "eager"), (8, "persistent_fwd"), (16, "persistent_bwd"), (32, "tile_pairs"),
               (64, "tile_rounds"), (128, "widened"), (256, "hub_geometry"), (512, "own_first"))


class _OdePlan:
    """A device-resident solve + discrete adjoint chosen by the library's ONE create call (ngpde_ode_create: include/ngpde.h, csrc/api_ode.hip):
    the right-hand side is described, the library checks that its layers chain and picks the plan -- a GAT-style layer (ngpde_node_gat_*),
    VMHConv(phi, gamma) (ngpde_node_vmh_*) or the two-GCNConv chain (ngpde_node_gcn2_*).  Holds the tape of ONE solve."""

    def __init__(self, handle, desc, kind):
        self.lib = _lib.load()
        _lib.flush_destroy()
        self.handle = handle
        self.ptr = None
        self.gen = 0
        self.kind = kind
        self.members = int(desc.members)
        self.n_nodes = int(handle._n_nodes)
        self.n_first, self.n_steps = int(desc.n_phi), int(desc.n_steps)
        out, fl = C.c_void_p(), C.c_int32()
        _lib.check(self.lib.ngpde_ode_create(handle.ptr, C.byref(desc), C.byref(out), C.byref(fl)))
        self.ptr, self._flags = out, fl.value

    def tape_bytes(self):
        return int(self.lib.ngpde_ode_tape_bytes(self.ptr))

    def flags(self):
        return {name for bit, name in _FLAG_NAMES if self._flags & bit} | {self.kind}

    def fault(self):
        f = C.c_int32()
        _lib.check(self.lib.ngpde_ode_fault(self.ptr, _lib.current_stream(), C.byref(f)))
        return bool(f.value)

    def claim(self):
        self._token = _Token()
        self._token_ref = weakref.ref(self._token)
        token, self._token = self._token, None
        self._pending = True
        return token

    def busy(self):
        ref = getattr(self, "_token_ref", None)
        return ref is not None and ref() is not None and getattr(self, "_pending", False)

    def __del__(self):
        try:
            if self.ptr:
                _lib.destroy_later("ngpde_ode_destroy", self.ptr)
                self.ptr = None
        except Exception:
            pass


def _ode_desc(rhs, tableau, n_steps, dt, with_backward, members=1, **kw):
    d = _lib.OdeDesc()
    d.rhs, d.tableau, d.n_steps, d.dt, d.with_backward, d.members = rhs, _lib.TABLEAU[tableau], int(n_steps), float(dt), int(with_backward), int(members)
    for k, v in kw.items():
        if k in ("phi_dims", "phi_acts", "gamma_dims", "gamma_acts"):
            for j, t in enumerate(v):
                getattr(d, k)[j] = int(t)
        else:
            setattr(d, k, v)
    return d


class _NodeOdeFn(torch.autograd.Function):
    """u(T) -- or with saveat the [T][N] array of the saved states -- of the plan's solve (ngpde_ode_forward), its pullback the discrete
    adjoint (ngpde_ode_backward).  args: u, plan, (save_every, save_start) or None, the attention vector or None, then per layer (weight
    [in][out], bias or None): the first `plan.n_first` layers are the descriptor's `first` stack (phi / the GAT weight / layer_1, layer_2),
    the rest its `second` (gamma)."""

    @staticmethod
    def forward(ctx, u, plan, save, att, *wb):
        lib = _lib.load()
        u = u.contiguous()
        ws = [w.contiguous() for w in wb[0::2]]
        bs = [None if b is None else b.contiguous() for b in wb[1::2]]
        att = None if att is None else att.contiguous()
        nf = plan.n_first if plan.n_first else len(ws)
        prm = _lib.OdeParams()
        for l, (w, b) in enumerate(zip(ws, bs)):
            blk, j = (prm.first, l) if l < nf else (prm.second, l - nf)
            blk.weight[j], blk.bias[j] = w.data_ptr(), (b.data_ptr() if b is not None else None)
        prm.attention = att.data_ptr() if att is not None else None
        k, start = save if save is not None else (0, 0)
        out = torch.empty_like(u) if save is None else torch.empty((plan.n_steps // k + int(start), u.numel()), dtype=torch.float32, device=u.device)
        _lib.check(lib.ngpde_ode_forward(plan.ptr, _lib.ptr(u), C.byref(prm), int(k), int(start), _lib.ptr(out), _lib.current_stream()))
        plan.gen += 1
        ctx.plan, ctx.gen, ctx.token, ctx.save, ctx.nf = plan, plan.gen, plan.claim(), (int(k), int(start)), nf
        ctx.save_for_backward(*([att] if att is not None else []), *ws)
        ctx.has_att, ctx.has_bias, ctx.ushape = att is not None, [b is not None for b in bs], u.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        plan = ctx.plan
        if ctx.gen != plan.gen:
            raise _lib.NgpdeError(_lib.ERR_STATE, "NeuralODE: another forward solve has replaced this solve's tape")
        saved = list(ctx.saved_tensors)
        att = saved.pop(0) if ctx.has_att else None
        ws = saved
        dev = ws[0].device
        dout = dout.contiguous()
        du0 = torch.empty(ctx.ushape, dtype=torch.float32, device=dev)
        dws = [torch.empty_like(w) for w in ws]
        dbs = [torch.empty((w.shape[1],), dtype=torch.float32, device=dev) if hb else None for w, hb in zip(ws, ctx.has_bias)]
        datt = torch.empty_like(att) if att is not None else None
        prm, gr = _lib.OdeParams(), _lib.OdeGrads()
        for l, (w, dw, db) in enumerate(zip(ws, dws, dbs)):
            (blk, gb), j = ((prm.first, gr.first), l) if l < ctx.nf else ((prm.second, gr.second), l - ctx.nf)
            blk.weight[j] = w.data_ptr()
            gb.dweight[j], gb.dbias[j] = dw.data_ptr(), (db.data_ptr() if db is not None else None)
        prm.attention = att.data_ptr() if att is not None else None
        gr.dattention = datt.data_ptr() if datt is not None else None
        _lib.check(lib.ngpde_ode_backward(plan.ptr, C.byref(prm), ctx.save[0], ctx.save[1], _lib.ptr(dout), _lib.ptr(du0), C.byref(gr),
                                          _lib.current_stream()))
        plan._pending = False
        grads = []
        for dw, db in zip(dws, dbs):
            grads += [dw, db]
        return (du0, None, None, datt, *grads)

"""NeuralODE -- the caller of the hot path in the reference's tutorial
(/root/reference/docs/src/tutorials/graph_node.md:44-66): `dudt(u, p, t) = model(u, p, st)` integrated
by an explicit Runge-Kutta scheme.  BASELINE configs fix the step count (Euler x 10, Tsit5 x 50), so
this integrator is fixed-step and its pullback is the discrete adjoint.

When the right-hand side is Chain(GCNConv(d => d, act), GCNConv(d => d, act)) on one graph (the
tutorial's `node_chain`, graph_node.md:78) the whole solve and its adjoint run device-resident from
HIP graphs (ngpde_node_gcn2_*).  Any other right-hand side (a GAT-style layer, VMHConv as in
docs/src/tutorials/VMH.md:85-89, ...) is stepped stage by stage through the layers' own kernels with
every Runge-Kutta combination -- forward and in the discrete adjoint -- as one ngpde_rk_stage_combine
launch (_NodeGenericFn): no torch element-wise kernel on the path.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

import torch

from . import _lib
from .graphs import GNNGraph
from .layers import AbstractExplicitLayer, Chain, GCNConv, rows_of
from .layers_mp import GATConv, VMHConv, _dense_stack, _node_data, _wt_b

_TSIT5_A = [
    [],
    [0.161],
    [-0.008480655492356989, 0.335480655492357],
    [2.8971530571054935, -6.359448489975075, 4.3622954328695815],
    [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525],
    [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383],
]
_TSIT5_B = [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
            2.324710524099774]
TABLEAUS = {"euler": ([[]], [1.0]), "tsit5": (_TSIT5_A, _TSIT5_B)}


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


def _padded_batch(g, device):
    """(padded graph, index of the real nodes in it) for a batch of single graphs whose sizes are not all multiples of the 32-row tile,
    cached on the batch; None when `g` is no such batch.  Every member keeps its node order and gets isolated nodes behind it up to a
    whole number of tiles; node data are zero there."""
    cached = getattr(g, "_vmh_pad", None)
    if cached is not None:
        return cached if cached[1].device == torch.device(device) else (cached[0], cached[1].to(device))
    members = getattr(g, "_members", None)
    if not members or list(g.ndata) != ["x"] or all(mg.num_nodes % 32 == 0 for mg in members):
        return None
    from .graphs import GNNGraph, _as_matrix_t
    sizes = np.array([mg.num_nodes for mg in members], dtype=np.int64)
    padded = (sizes + 31) // 32 * 32
    off, poff = np.concatenate([[0], np.cumsum(sizes)]), np.concatenate([[0], np.cumsum(padded)])
    index = np.concatenate([np.arange(n, dtype=np.int64) + po for n, po in zip(sizes, poff[:-1])])
    s0, t0 = g.edge_index(index_base=0)
    s0, t0 = np.asarray(s0.cpu() if isinstance(s0, torch.Tensor) else s0), np.asarray(t0.cpu() if isinstance(t0, torch.Tensor) else t0)
    gp = GNNGraph(index[s0], index[t0], num_nodes=int(poff[-1]), index_base=0, num_graphs=len(members))
    x = _as_matrix_t(g.ndata["x"], g.num_nodes).to(device)                      # [N][pd]
    idx_t = torch.as_tensor(index, device=device)
    xp = torch.zeros((int(poff[-1]), x.shape[1]), dtype=torch.float32, device=device).index_copy(0, idx_t, x.to(torch.float32))
    gp.ndata = {"x": xp.T}
    order = g._shared.get("order")
    if order is not None:      # the members' locality orders, each followed by its padding nodes
        parts = []
        for k in range(len(members)):
            parts.append(index[np.asarray(order[off[k]:off[k + 1]], dtype=np.int64)])
            parts.append(np.arange(poff[k] + sizes[k], poff[k + 1], dtype=np.int64))
        gp._shared["order"] = np.concatenate(parts).astype(np.int32)
    g._vmh_pad = (gp, idx_t)
    return g._vmh_pad


_CANON_BATCHES = {}      # sorted member ids -> (the first batch seen of these members, its members): at most _CANON_MAX entries
_CANON_MAX = 2


def _canonical_batch(g, device):
    """A DataLoader(shuffle = true) hands the training loop the SAME point clouds in a new order every epoch (VMH.md:120-134): a new
    block-diagonal graph whose members are the members of an earlier batch, permuted.  The trajectories of a batch's members are
    independent, so such a batch is solved on the earlier batch's graph -- its handle, plan and tapes -- with the state's rows sent
    through the permutation.  Returns (earlier batch, int64 map: node of `g` -> node of the earlier batch) or None (`g` is no batch
    of single graphs, or the first of its kind: it is remembered).  Members are compared by identity; the entry keeps them alive."""
    members = getattr(g, "_members", None)
    if not members or len(members) < 2 or list(g.ndata) != ["x"] or os.environ.get("NGPDE_NO_BATCH_REUSE") == "1":
        return None
    # identity of the members AND of their positions' storage (data pointer + in-place version counter): a member whose cloud was
    # moved in place, or whose ndata["x"] was reassigned, is another cloud -- it must not be solved with the first batch's positions
    def stamp(mg):
        x = mg.ndata.get("x") if isinstance(mg.ndata, dict) else None
        return (id(mg), x.data_ptr(), x._version) if isinstance(x, torch.Tensor) else (id(mg), id(x), 0)
    key = tuple(sorted(stamp(mg) for mg in members))
    hit = _CANON_BATCHES.get(key)
    if hit is None:
        _CANON_BATCHES[key] = (g, list(members))
        while len(_CANON_BATCHES) > _CANON_MAX:
            _CANON_BATCHES.pop(next(iter(_CANON_BATCHES)))
        return None
    g0, members0 = hit
    _CANON_BATCHES[key] = _CANON_BATCHES.pop(key)
    if g0 is g:
        return None
    cached = getattr(g, "_vmh_canon", None)
    if cached is not None and cached[0] is g0:
        return cached if cached[1].device == torch.device(device) else (g0, cached[1].to(device))
    off0, slots = np.concatenate([[0], np.cumsum([mg.num_nodes for mg in members0])]), {}
    for j, mg in enumerate(members0):
        slots.setdefault(id(mg), []).append(j)          # (a cloud that occurs twice: its copies are interchangeable)
    parts = [np.arange(mg.num_nodes, dtype=np.int64) + off0[slots[id(mg)].pop()] for mg in members]
    g._vmh_canon = (g0, torch.as_tensor(np.concatenate(parts), device=device))
    return g._vmh_canon


# ---- any right-hand side: explicit RK stepping with every combination as ONE library launch ---------------------------------


def _combine(base, c_self, terms, coefs, out=None):
    """out = c_self * base + sum_k coefs[k] * terms[k]  (ngpde_rk_stage_combine; row-major [N][D] float32 tensors of one size)"""
    lib = _lib.load()
    ref = base if base is not None else terms[0]
    if out is None:
        out = torch.empty_like(ref, memory_format=torch.contiguous_format)
    arr = (C.c_void_p * max(len(terms), 1))(*[t.data_ptr() for t in terms])
    cf = (C.c_float * max(len(coefs), 1))(*[float(c) for c in coefs])
    _lib.check(lib.ngpde_rk_stage_combine(ref.numel(), float(c_self), _lib.ptr(base), len(terms), arr, cf, _lib.ptr(out),
                                          _lib.current_stream()))
    return out


def _dense(t):
    """a view of `t` whose memory order is its index order (the cotangent of a (out x in) weight arrives as the transpose of
    the kernels' [in][out] array): element-wise combinations then need no copy"""
    if t.is_contiguous():
        return t
    if t.dim() == 2 and t.T.is_contiguous():
        return t.T
    return t.contiguous()


def _leaves(tree):
    out = []
    for v in tree.values():
        if isinstance(v, dict):
            out += _leaves(v)
        else:
            out.append(v)
    return out


def _graph_leaves(st):
    """the GNNGraph leaves of a state tree, in traversal order (the leaves updategraph replaces, src/utils.jl:24-31)"""
    out = []
    if isinstance(st, dict):
        for v in st.values():
            if isinstance(v, GNNGraph):
                out.append(v)
            elif isinstance(v, dict):
                out += _graph_leaves(v)
    return out


def _rebuild(tree, it):
    return {k: (_rebuild(v, it) if isinstance(v, dict) else next(it)) for k, v in tree.items()}


def _rk_forward(node, u, ps_in, st, needs, fresh=False):
    """u(T) of the fixed-step solve; with `needs` also the tape: per step and stage the (stage input, stage output) pair whose
    autograd closure is the layers' pullback.  fresh: the stage inputs get version counters of their own (a captured solve
    refills its static input buffer before every replay; the closures are only ever run at capture time)"""
    a, b = TABLEAUS[node.solver]
    dt, S = node.dt, len(b)
    ucur, tape, st_out = u, [], st
    k_save = node.save_every
    saves = [ucur] if (k_save and node.save_start) else []      # saveat: the states at t0 (+ j saveat), as DiffEq's sol.u
    with (torch.enable_grad() if needs else torch.no_grad()):
        for n_step in range(node.n_steps):
            ks, pairs = [], []
            for i in range(S):
                terms = [ks[j] for j in range(i) if a[i][j] != 0.0]
                coefs = [dt * a[i][j] for j in range(i) if a[i][j] != 0.0]
                U = _combine(ucur, 1.0, terms, coefs) if terms else ucur
                if needs:
                    U = (U.data if fresh else U.detach()).requires_grad_(True)
                k, st_out = node.model(U.T, ps_in, st_out)
                k = rows_of(k)
                if k.shape != ucur.shape:
                    raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                                 f"DimensionMismatch: the right-hand side maps {tuple(ucur.shape[::-1])} to "
                                                 f"{tuple(k.shape[::-1])}; du/dt must have the shape of u")
                ks.append(k.detach())
                pairs.append((U, k))
            ucur = _combine(ucur, 1.0, ks, [dt * bi for bi in b])
            if needs:
                tape.append(pairs)
            if k_save and (n_step + 1) % k_save == 0:
                saves.append(ucur)
    if k_save:      # [T][N][D]: the memory layout of the reference's (D x N x T) array; rows written by library launches
        out = torch.empty((len(saves),) + tuple(ucur.shape), dtype=ucur.dtype, device=ucur.device)
        for j, s_ in enumerate(saves):
            _combine(s_, 1.0, [], [], out=out[j])
        ucur = out
    return ucur, tape, st_out


def _accumulate_many(pairs):
    """acc += g for every (acc, g) pair of dense float32 tensors of equal size (ngpde_accumulate_many: one launch per 24 arrays)"""
    if not pairs:
        return
    for a_, g_ in pairs:
        if a_.numel() != g_.numel() or a_.dtype != torch.float32 or g_.dtype != torch.float32:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, "parameter cotangent and its accumulator differ in size or type")
    n = len(pairs)
    accs = (C.c_void_p * n)(*[a_.data_ptr() for a_, _ in pairs])
    gs = (C.c_void_p * n)(*[g_.data_ptr() for _, g_ in pairs])
    cnt = (C.c_int64 * n)(*[a_.numel() for a_, _ in pairs])
    _lib.check(_lib.load().ngpde_accumulate_many(n, accs, gs, cnt, _lib.current_stream()))


def _rk_backward(node, tape, duT, params, retain=False):
    """discrete adjoint of _rk_forward: (du0, cotangents of `params`)"""
    a, b = TABLEAUS[node.solver]
    dt, S = node.dt, len(b)
    acc = [None] * len(params)
    k_save = node.save_every
    off = 0 if node.save_start else -1           # saved state j is u after (j - off') steps: index of u_n is n / k_save + off
    lam = duT[node.n_steps // k_save + off] if k_save else duT
    for n_step, pairs in zip(range(len(tape) - 1, -1, -1), reversed(tape)):
        ubar = [None] * S
        for i in reversed(range(S)):
            terms = [ubar[j] for j in range(i + 1, S) if a[j][i] != 0.0 and ubar[j] is not None]
            coefs = [dt * a[j][i] for j in range(i + 1, S) if a[j][i] != 0.0 and ubar[j] is not None]
            if b[i] == 0.0 and not terms:
                continue
            kbar = _combine(lam, dt * b[i], terms, coefs)
            U, k = pairs[i]
            grads = torch.autograd.grad(k, [U] + params, kbar, allow_unused=True, retain_graph=retain)
            if grads[0] is not None:
                ubar[i] = grads[0].contiguous()
            pairs_ag = []
            for n, g in enumerate(grads[1:]):
                if g is None:
                    continue
                if acc[n] is None:
                    # own the accumulator in a layout _dense can view without a copy (an expanded / strided cotangent would make
                    # _dense return a temporary, and the sum written into it would be lost)
                    acc[n] = g if (g.is_contiguous() or (g.dim() == 2 and g.T.is_contiguous())) else g.contiguous()
                elif acc[n].stride() == g.stride():
                    # same layout for every stage's cotangent of one parameter: add in memory order
                    pairs_ag.append((_dense(acc[n]), _dense(g)))
                else:       # layouts differ (one transposed, one not): index-wise sum in the accumulator's own layout
                    pairs_ag.append((_dense(acc[n]), _dense(g.contiguous() if acc[n].is_contiguous() else g.T.contiguous().T)))
            _accumulate_many(pairs_ag)      # ONE launch for all parameters of this stage (sixteen arrays in the VMH tutorial's model)
        live = [x for x in ubar if x is not None]
        if k_save and n_step % k_save == 0 and n_step // k_save + off >= 0:
            live.append(duT[n_step // k_save + off])          # lambda(t_n) also carries the cotangent of the state saved there
        if live:
            lam = _combine(lam, 1.0, live, [1.0] * len(live))
    return lam, acc


def _inner_params(ps, leaves, fresh=False):
    """aliases of the parameters that are leaves of the stages' autograd closures, and the tree holding them (fresh: with
    version counters of their own, so that an optimiser's in-place update between the captures does not invalidate them)"""
    inner = [(p.data if fresh else p.detach()).requires_grad_(p.requires_grad) if isinstance(p, torch.Tensor) else p for p in leaves]
    return inner, _rebuild(ps, iter(inner))


class _NodeGenericFn(torch.autograd.Function):
    """Fixed-step explicit Runge-Kutta solve of du/dt = model(u) for ANY model made of this package's layers, and its discrete
    adjoint.  Forward: the stage inputs u + dt sum_j a_ij k_j and the step update are one ngpde_rk_stage_combine launch each,
    the stages are the layers' own kernels; every stage keeps its (input, output) pair with the layer's pullback closure.
    Backward: per stage, in reverse, K-bar_i = dt b_i lambda + dt sum_{j>i} a_ji U-bar_j (one launch), U-bar_i and the parameter
    cotangents from the stage's pullback (the layers' backward kernels), parameter gradients accumulated by one launch each,
    lambda += sum_j U-bar_j (one launch).  No torch element-wise kernel takes part.
    [docs/src/tutorials/VMH.md:85-89 NeuralODE(VMHConv); graph_node.md:44-66; BASELINE config 3 "GAT as ODE RHS"]"""

    @staticmethod
    def forward(ctx, u, node, ps, st, *leaves):
        needs = any(ctx.needs_input_grad)
        inner, ps_in = _inner_params(ps, leaves) if needs else (list(leaves), ps)
        uT, tape, _ = _rk_forward(node, u.detach().contiguous(), ps_in, st, needs)
        ctx.node, ctx.tape, ctx.inner = node, tape if needs else None, inner
        return uT

    @staticmethod
    def backward(ctx, duT):
        params = [p for p in ctx.inner if isinstance(p, torch.Tensor) and p.requires_grad]
        lam, acc = _rk_backward(ctx.node, ctx.tape, duT.contiguous(), params)
        ctx.tape = None
        it = iter(acc)
        out = [next(it) if (isinstance(p, torch.Tensor) and p.requires_grad) else None for p in ctx.inner]
        return (lam, None, None, None, *out)


class _NoCyclicGC:
    """No cyclic garbage collection while a HIP graph is being captured: a collection in the middle of a capture may finalise an older
    captured solve, and destroying ITS graphs is an operation the capturing stream does not permit (the capture dies with
    hipErrorStreamCaptureUnsupported).  torch.cuda.graph collects once before the capture starts; this keeps it that way until it ends."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


class _CapturedSolve:
    """The generic solve as two HIP graphs (NeuralODE(..., capture=True)): the forward stepping loop -- every layer launch
    and every Runge-Kutta combination of all steps -- is captured once and replayed per call, and so is the whole discrete
    adjoint the first time a backward pass asks for it.  Static buffers: the input is copied in, u(T) and the cotangents are
    the graphs' own buffers (valid until the next call of the same NeuralODE on the same arguments).  The captures bake in the
    addresses of the parameters and of the graph's derived arrays: a call with other parameter tensors, another graph or
    another input shape captures anew (NeuralODE keeps the most recent few)."""

    def __init__(self, node, u, ps, st, leaves, needs):
        self.node, self.needs = node, needs
        self.leaves = list(leaves)                         # keeps the captured addresses alive
        self.inner, self.ps_in = _inner_params(ps, leaves, fresh=True) if needs else (list(leaves), ps)
        self.params = [p for p in self.inner if isinstance(p, torch.Tensor) and p.requires_grad] if needs else []
        self.u_static = u.detach().clone()
        self.st = st
        # one eager solve on a side stream first: graph handles, workspaces and lazily built tables exist before the capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            _rk_forward(node, self.u_static, self.ps_in, st, False)
        torch.cuda.current_stream().wait_stream(side)
        self.fwd_graph = torch.cuda.CUDAGraph()
        # (relaxed: a finaliser that frees device memory in the middle of the capture must not invalidate it)
        with _NoCyclicGC(), torch.cuda.graph(self.fwd_graph, capture_error_mode="relaxed"):
            self.uT_static, self.tape, _ = _rk_forward(node, self.u_static, self.ps_in, st, needs, fresh=True)
        self.bwd_graph = None
        self.generation = 0                                # forward replays so far: the single static tape belongs to the last one

    def forward(self, u):
        self.u_static.copy_(u)
        self.fwd_graph.replay()
        self.generation += 1
        return self.uT_static

    def backward(self, duT, generation):
        if generation != self.generation:
            # `y1 = node(u1); y2 = node(u2); (y1 + y2).backward()`: the second replay has overwritten the first solve's tape
            raise _lib.NgpdeError(_lib.ERR_STATE, "NeuralODE(capture=True): another forward solve has replaced this solve's tape "
                                                  "(a captured solve holds ONE tape); run each backward before the next forward "
                                                  "on the same arguments, or use capture=False")
        if self.bwd_graph is None:
            self.duT_static = duT.detach().clone()
            self.bwd_graph = torch.cuda.CUDAGraph()
            with _NoCyclicGC(), torch.cuda.graph(self.bwd_graph, pool=self.fwd_graph.pool(), capture_error_mode="relaxed"):
                self.lam_static, self.acc_static = _rk_backward(self.node, self.tape, self.duT_static, self.params, retain=True)
        else:
            self.duT_static.copy_(duT)
        self.bwd_graph.replay()
        return self.lam_static, self.acc_static


class _NodeCapturedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, solve, *leaves):
        ctx.solve = solve
        uT = solve.forward(u.detach().contiguous()).clone()
        ctx.generation = solve.generation
        return uT

    @staticmethod
    def backward(ctx, duT):
        lam, acc = ctx.solve.backward(duT.contiguous(), ctx.generation)
        it = iter(acc)
        out = [next(it) if (isinstance(p, torch.Tensor) and p.requires_grad) else None for p in ctx.solve.inner]
        return (lam.clone(), None, *[None if g is None else g.clone() for g in out])


class NeuralODE(AbstractExplicitLayer):
    """NeuralODE(model; solver="tsit5", tspan=(0, 1), n_steps=..., dt=None)

    A Lux container with the single field `model`, so `ps` and `st` are the model's own
    (graph_node.md:44-52, :59-66).  `solver` is "euler" or "tsit5"; the step is fixed:
    dt = (tspan[1] - tspan[0]) / n_steps unless given.  capture=True (an extension, for right-hand sides other than the
    two-GCNConv chain, which is always device-resident): the whole solve and its adjoint are captured into HIP graphs at
    the first call and replayed afterwards (_CapturedSolve).
    """

    def __init__(self, model, *, solver="tsit5", tspan=(0.0, 1.0), n_steps=10, dt=None, capture=False, saveat=None, save_start=True):
        solver = solver.lower()
        if solver not in TABLEAUS:
            raise _lib.ArgumentError(_lib.ERR_INVALID_ARGUMENT, f"unknown solver {solver!r}; one of {list(TABLEAUS)}")
        self.model, self.solver, self.tspan, self.n_steps = model, solver, tuple(tspan), int(n_steps)
        self.dt = float(dt) if dt is not None else (self.tspan[1] - self.tspan[0]) / self.n_steps
        # saveat (VMH.md:85 `NeuralODE(gnn, tspan, Tsit5(); saveat=dt_train)`): the output is the solution at t0, t0 + saveat, ..., T --
        # a (D x N x T) array -- instead of u(T).  The step is fixed, so saveat must be a whole number of steps.
        self.save_every, self.save_start = 0, bool(save_start)
        if saveat is not None:
            k = round(float(saveat) / self.dt)
            if k < 1 or abs(k * self.dt - float(saveat)) > 1e-6 * max(abs(float(saveat)), 1.0) or self.n_steps % k:
                raise _lib.ArgumentError(_lib.ERR_INVALID_ARGUMENT, f"NeuralODE: saveat = {saveat} must be a whole number of steps "
                                                                  f"(dt = {self.dt}) that divides n_steps = {self.n_steps}")
            self.save_every = k
        self._plans = {}
        self._no_member_plan = False
        self.capture = bool(capture)      # generic right-hand sides: replay the whole solve / adjoint from HIP graphs
        self._captured = {}
        self._gat_ok = {}                 # (id(graph handle), heads) -> (handle, does the device-resident GAT solver take it?)

    def initialparameters(self, rng):
        return self.model.initialparameters(rng)

    def initialstates(self, rng):
        return self.model.initialstates(rng)

    def statelength(self):
        return self.model.statelength()

    # -- is the right-hand side the tutorial's two-GCNConv chain on one graph? ---------------------
    max_plans = 2   # cached solver-plan keys per NeuralODE (forward-only and forward+backward of the current graph)
    max_outstanding = 8   # plans per key: one per solve whose backward is still outstanding (each owns a tape)

    def _gcn2(self, ps, st):
        m = self.model
        if not (isinstance(m, Chain) and len(m.chain) == 2 and all(isinstance(l, GCNConv) for l in m.chain)):
            return None
        l1, l2 = m.chain
        d = l1.in_chs
        same = (l1.out_chs == d and l2.in_chs == d and l2.out_chs == d and l1.act == l2.act
                and l1.add_self_loops == l2.add_self_loops and l1.use_edge_weight == l2.use_edge_weight
                and l1.bias == l2.bias)
        if not same or d not in (16, 32, 64, 128):
            return None
        g1, g2 = st["layer_1"]["graph"], st["layer_2"]["graph"]
        # copies made by wrapgraph / updategraph share one handle cache: same structure without comparing arrays
        if g1 is not g2 and g1._handles is not g2._handles and g1 != g2:
            return None
        if l1.use_edge_weight and (g1.edge_weight is None or g2.edge_weight is not g1.edge_weight):
            return None      # (the generic solver raises the layer's own error / handles two weight vectors)
        return l1, g1, d

    def plan_for(self, ps, st, with_backward):
        info = self._gcn2(ps, st)
        if info is None:
            return None
        l1, g, d = info
        # use_edge_weight=true: the graph's stored weights in the messages, the unweighted degree in the normalisation
        # (src/layers.jl:224 vs :230, as GCNConv.__call__ does); the plan's kernels read them from the handle's slot lists
        norm = (l1.add_self_loops, g.edge_weight if l1.use_edge_weight else None, False)
        # a batch of graphs that share ONE structure (batch([g, g, ...]) or copies of g with other features): the plan is built on
        # the member and solves the trajectories one after the other inside its persistent launches
        members = getattr(g, "_members", None)
        member_plan = (members is not None and len(members) > 1 and all(m._handles is members[0]._handles for m in members)
                       and not self._no_member_plan and not l1.use_edge_weight)
        handle = members[0].handle(norm) if member_plan else g.handle(norm)
        key = (id(handle), d, l1.act, bool(with_backward), len(members) if member_plan else 1)
        pool = self._plans.get(key)
        if pool is None:
            pool = self._plans[key] = []
            while len(self._plans) > self.max_plans:        # a plan owns its tape (GBs): keep only the most recent ones
                self._plans.pop(next(iter(self._plans)))     # (a training loop that swaps the graph every minibatch)
        else:
            self._plans[key] = self._plans.pop(key)          # most recently used last
        # a plan holds ONE solve's tape: `y1 = node(u1); y2 = node(u2); (y1 + y2).backward()` needs two
        for plan in pool:
            if not (with_backward and plan.busy()):
                return plan
        if len(pool) >= self.max_outstanding:
            raise _lib.NgpdeError(_lib.ERR_STATE, f"NeuralODE: {len(pool)} solves await their backward pass on this graph; "
                                                  "each holds a tape -- run backward (or raise NeuralODE.max_outstanding)")
        try:
            plan = _Plan(handle, d, l1.act, self.solver, self.n_steps, self.dt, with_backward,
                         members=len(members) if member_plan else 1)
        except _lib.NgpdeError as e:
            if not (member_plan and e.code == _lib.ERR_UNSUPPORTED):
                raise
            self._no_member_plan = True     # not a case of the persistent plan: one handle for the whole batch instead
            self._plans.pop(key, None)
            return self.plan_for(ps, st, with_backward)
        pool.append(plan)
        return plan

    def gat_plan_for(self, ps, st, u):
        """the device-resident plan when the right-hand side is ONE GAT-style layer in the one-launch shape (64 => heads x c = 64,
        concat, tiles fit the LDS halo) and the library takes it (ngpde_node_gat_supported); None otherwise"""
        m = self.model
        if isinstance(m, Chain) and len(m.chain) == 1 and isinstance(m.chain[0], GATConv):    # Chain(GATConv(...)): the same solve
            m, ps, st = m.chain[0], ps["layer_1"], st["layer_1"]
        if not (isinstance(m, GATConv) and m.concat and u.is_cuda and m.in_chs == 64 and m.heads * m.out_chs == 64):
            return None
        g = st["graph"]
        # a batch of graphs that share ONE structure: the plan is built on the member, two members per workgroup
        members = getattr(g, "_members", None)
        member_plan = members is not None and len(members) > 1 and all(mm._handles is members[0]._handles for mm in members)
        handle = m._graph(members[0] if member_plan else g).handle()
        with_backward = torch.is_grad_enabled() and (u.requires_grad or any(
            isinstance(v, torch.Tensor) and v.requires_grad for v in ps.values()))
        key = ("gat", id(handle), m.heads, m.act, m.negative_slope, bool(with_backward), len(members) if member_plan else 1)
        pool = self._plans.get(key)
        if pool is None:
            # asked once per (graph handle, head count) of this NeuralODE: the check compares the two directions' schedules on the
            # device and synchronises (the entry keeps the handle alive, so its id cannot be recycled)
            ok = self._gat_ok.get((id(handle), m.heads))
            if ok is None:
                ok = (handle, bool(_lib.load().ngpde_node_gat_supported(handle.ptr, 64, m.heads, m.out_chs)))
                self._gat_ok[(id(handle), m.heads)] = ok
                while len(self._gat_ok) > 8:
                    self._gat_ok.pop(next(iter(self._gat_ok)))
            if not ok[1]:
                return None
            pool = self._plans[key] = []
            while len(self._plans) > self.max_plans:
                self._plans.pop(next(iter(self._plans)))
        else:
            self._plans[key] = self._plans.pop(key)
        for plan in pool:
            if not (with_backward and plan.busy()):
                return plan
        if len(pool) >= self.max_outstanding:
            raise _lib.NgpdeError(_lib.ERR_STATE, f"NeuralODE: {len(pool)} solves await their backward pass on this graph; "
                                                  "each holds a tape -- run backward (or raise NeuralODE.max_outstanding)")
        plan = _OdePlan(handle, _ode_desc(_lib.RHS_GAT, self.solver, self.n_steps, self.dt, with_backward, len(members) if member_plan else 1, width=64,
                                          act=int(m.act), heads=int(m.heads), head_width=int(m.out_chs), negative_slope=float(m.negative_slope)), "gat")
        pool.append(plan)
        return plan

    def vmh_plan_for(self, ps, st, u, needs_grad):
        """the device-resident plan when the right-hand side is VMHConv(phi, gamma) on a scalar state in the shapes the library takes
        (ngpde_node_vmh_supported: docs/src/tutorials/VMH.md:75-89's model); (plan, weights and biases) or None"""
        m = self.model
        if not (isinstance(m, VMHConv) and u.is_cuda and u.dim() == 2 and u.shape[1] == 1):
            return None
        g = st["graph"]
        if list(g.ndata) != ["x"]:      # (a batched graph is one block-diagonal graph to this plan: its members' tiles never neighbour)
            return None
        remap = _canonical_batch(g, u.device)      # the members of an earlier batch in a new order: that batch's graph and plan serve
        nodemap, g_given = None, g
        if remap is not None:
            g, nodemap = remap
        try:
            phi, gam = _dense_stack(m.ϕ, ps["ϕ"], "ϕ"), _dense_stack(m.γ, ps["γ"], "γ")
        except _lib.NgpdeError:
            return None
        pos = _node_data(g, u.device)
        pd = pos.shape[1]
        wb, dims, acts = [], [], []
        for stack in (phi, gam):
            d, a = [], []
            for layer, p in stack:
                wt, b = _wt_b(p)
                wb += [wt, b]
                d.append(wt.shape[0])
                a.append(layer.act)
            d.append(rows_of(stack[-1][1]["weight"]).shape[1])
            dims.append(d)
            acts.append(a)
        aggr = _lib.AGGR.get(m.aggr)
        if aggr is None:
            return None
        # the plan's entries take pointers only: a parameter tree that does not chain (layer l's output width against layer l + 1's
        # input width, every bias against its layer, phi's input = [h_i; h_j - h_i; x_j - x_i], gamma's = [h_i; m_i]) would make the
        # kernels read past the weight arrays -- the reference fails in the matrix product with a DimensionMismatch
        k = 0
        for name, d in (("ϕ", dims[0]), ("γ", dims[1])):      # (that the stacks chain as the layer feeds them is ngpde_ode_create's check)
            _check_plan_shapes("NeuralODE(VMHConv)", u, u.shape[0], 1,
                               [(f"{name}.layer_{l + 1}.weight", wb[k + 2 * l], (d[l], d[l + 1])) for l in range(len(d) - 1)],
                               [(f"{name}.layer_{l + 1}.bias", wb[k + 2 * l + 1], d[l + 1]) for l in range(len(d) - 1)])
            k += 2 * (len(d) - 1)
        lib = _lib.load()
        ia = lambda v: (C.c_int32 * len(v))(*[int(t) for t in v])
        supported = lambda h: lib.ngpde_node_vmh_supported(h.ptr, 1, pd, len(acts[0]), ia(dims[0]), ia(acts[0]), len(acts[1]), ia(dims[1]),
                                                            ia(acts[1]), aggr)
        members = getattr(g, "_members", None)
        straddles = bool(members) and len(members) > 1 and any(mg.num_nodes % 32 for mg in members)
        index, pool, key, handle = None, None, None, None
        if not straddles:      # (a batch whose clouds share tiles goes to its padded form at once: no handle of the unpadded union is built)
            handle = g.handle()
            key = ("vmh", id(handle), tuple(dims[0]), tuple(acts[0]), tuple(dims[1]), tuple(acts[1]), aggr, bool(needs_grad))
            pool = self._plans.get(key)
        if pool is None and (straddles or not supported(handle)):
            # a batch of point clouds whose sizes are not multiples of the 32-row tile (VMH.md:120-134: 24 clouds of 3 000 points): a tile
            # that holds the end of one cloud and the start of the next stages two neighbourhoods and can overflow its halo, which takes
            # the persistent forms away from the whole handle.  The same batch with every cloud padded to whole tiles by isolated nodes
            # (no edges: no messages, a zero cotangent on their outputs) is served; u goes in and out through an index map.
            pad = _padded_batch(g, u.device)
            if pad is None:
                return None
            g, index = pad
            handle, pos = g.handle(), _node_data(g, u.device)
            key = ("vmh", id(handle), tuple(dims[0]), tuple(acts[0]), tuple(dims[1]), tuple(acts[1]), aggr, bool(needs_grad))
            pool = self._plans.get(key)
            if pool is None and not supported(handle):
                return None
        if pool is None:
            pool = self._plans[key] = []
            while len(self._plans) > self.max_plans:
                self._plans.pop(next(iter(self._plans)))
        else:
            self._plans[key] = self._plans.pop(key)
        if nodemap is not None:      # node of the given batch -> node of the earlier batch (-> its row among the padding nodes'), kept on the batch
            comp = getattr(g_given, "_vmh_comp", None)
            if comp is None or comp[0] is not index or comp[1] is not nodemap:
                comp = g_given._vmh_comp = (index, nodemap, nodemap if index is None else index.index_select(0, nodemap))
            index = comp[2]
        for plan in pool:
            if not (needs_grad and plan.busy()):
                return plan, wb, index
        if len(pool) >= self.max_outstanding:
            raise _lib.NgpdeError(_lib.ERR_STATE, f"NeuralODE: {len(pool)} solves await their backward pass on this graph; "
                                                  "each holds a tape -- run backward (or raise NeuralODE.max_outstanding)")
        # plans of this right-hand side on OTHER graphs are of no use any more (updategraph per minibatch, VMH.md:132-134): their tapes --
        # tens of GB at the tutorial's batch size -- go back to the library's pool before the new plan asks for its own
        # (only when THIS plan needs tapes, and only plans that hold tapes: a forward-only validation solve between training steps
        # must not evict the training graph's plan and make the next step rebuild it)
        if needs_grad:
            for old_key in [k for k in self._plans if k[0] == "vmh" and k[1] != id(handle) and k[-1]]:
                self._plans.pop(old_key)
        try:
            plan = _OdePlan(handle, _ode_desc(_lib.RHS_VMH, self.solver, self.n_steps, self.dt, needs_grad, width=1, pos_width=int(pd), aggr=int(aggr),
                                              pos=pos.data_ptr(), n_phi=len(acts[0]), phi_dims=dims[0], phi_acts=acts[0], n_gamma=len(acts[1]),
                                              gamma_dims=dims[1], gamma_acts=acts[1]), "vmh")
        except _lib.NgpdeError as e:
            # the tapes of a solve -- every layer's input rows and dz rows of every right-hand-side evaluation, 64 floats wide -- did not
            # fit the device (the library parks and re-uses the tapes of plans that went away, and frees them before it gives up): the
            # generic solver, which keeps O(stages) arrays, takes the solve
            if e.code != _lib.ERR_HIP:
                raise
            if not pool:
                self._plans.pop(key, None)
            return None
        pool.append(plan)
        return plan, wb, index

    def __call__(self, x, ps, st):
        u = rows_of(x)
        needs_grad = torch.is_grad_enabled() and (u.requires_grad or any(
            isinstance(v, torch.Tensor) and v.requires_grad for v in _leaves(ps)))      # (parameter trees of any depth: VMHConv's phi / gamma chains)
        plan = self.plan_for(ps, st, needs_grad) if not self.save_every else None
        if plan is not None:
            if not u.is_cuda:
                raise _lib.ArgumentError(_lib.ERR_INVALID_ARGUMENT, "NeuralODE: inputs must live on the GPU (no CPU fallback)")
            p1, p2 = ps["layer_1"], ps["layer_2"]
            b1 = p1["bias"].reshape(-1) if "bias" in p1 else None
            b2 = p2["bias"].reshape(-1) if "bias" in p2 else None
            w1, w2 = rows_of(p1["weight"]), rows_of(p2["weight"])
            # the plan's entries take no sizes (they walk the handle's rows): what GCNConv.__call__ would have checked
            # (check_num_nodes, the matrix product's own DimensionMismatch) is checked here, before any kernel runs
            d = self.model.chain[0].in_chs
            _check_plan_shapes("NeuralODE(GCNConv, GCNConv)", u, plan.n_nodes * plan.members, d,
                               [("layer_1.weight", w1, (d, d)), ("layer_2.weight", w2, (d, d))],
                               [("layer_1.bias", b1, d), ("layer_2.bias", b2, d)])
            uT = _NodeGCN2Fn.apply(u, w1, b1, w2, b2, plan)
            return uT.T, st
        gplan = self.gat_plan_for(ps, st, u) if not self.save_every else None
        if gplan is not None:
            gps = ps["layer_1"] if isinstance(self.model, Chain) else ps
            gm = self.model.chain[0] if isinstance(self.model, Chain) else self.model
            b = gps["bias"].reshape(-1) if "bias" in gps else None
            w, a = rows_of(gps["weight"]), rows_of(gps["a"])
            _check_plan_shapes("NeuralODE(GATConv)", u, gplan.n_nodes * gplan.members, 64, [("weight", w, (64, 64))], [("bias", b, 64)])
            if a.numel() != 2 * gm.out_chs * gm.heads:
                raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, f"DimensionMismatch: NeuralODE(GATConv): attention vector has "
                                             f"{a.numel()} entries, expected 2 x {gm.out_chs} x {gm.heads}")
            uT = _NodeOdeFn.apply(u, gplan, None, a, w, b)
            return uT.T, st
        vplan = self.vmh_plan_for(ps, st, u, needs_grad)
        if vplan is not None:
            plan_v, wb, index = vplan
            # a state whose node count is not the plan's graph's (a forgotten updategraph in the minibatch loop): the reference's
            # check_num_nodes DimensionMismatch, not N floats read and written through buffers of N'
            n_expected = plan_v.n_nodes if index is None else int(index.numel())
            if u.shape[0] != n_expected:
                raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, f"DimensionMismatch: NeuralODE(VMHConv): the state has {u.shape[0]} "
                                             f"nodes, the graph in the layer's state has {n_expected}")
            uin = u.reshape(-1)
            if index is not None:      # (a padded batch: the real nodes' rows among the isolated padding nodes')
                uin = _rows_index(uin, index, plan_v.n_nodes, True)
            if self.save_every:      # saveat: the (1 x N x T) array of the solution at t0 (+ j saveat)
                us = _NodeOdeFn.apply(uin, plan_v, (self.save_every, self.save_start), None, *wb)
                if index is not None:
                    us = _rows_index(us, index, plan_v.n_nodes, False)
                return us.T.unsqueeze(0), st
            uT = _NodeOdeFn.apply(uin, plan_v, None, None, *wb)
            if index is not None:
                uT = _rows_index(uT, index, plan_v.n_nodes, False)
            return uT.reshape(u.shape).T, st
        # any other right-hand side: explicit RK stepping through the layers' own kernels, every Runge-Kutta combination (and
        # every combination of the discrete adjoint) one library launch
        if not u.is_cuda:
            raise _lib.ArgumentError(_lib.ERR_INVALID_ARGUMENT, "NeuralODE: inputs must live on the GPU (no CPU fallback)")
        leaves = _leaves(ps)
        if self.capture:
            needs = torch.is_grad_enabled() and (u.requires_grad or any(isinstance(p, torch.Tensor) and p.requires_grad for p in leaves))
            # every GNNGraph leaf of the state tree (a container's graphs sit in st["layer_k"]["graph"]): the captures bake in
            # the addresses of their derived arrays.  The entry keeps `st` -- and with it the graphs -- alive, so an id cannot be
            # recycled while its key is cached.
            key = (tuple(u.shape), needs, tuple(id(g) for g in _graph_leaves(st)),
                   tuple(p.data_ptr() if isinstance(p, torch.Tensor) else id(p) for p in leaves))
            solve = self._captured.get(key)
            if solve is None:
                solve = self._captured[key] = _CapturedSolve(self, u, ps, st, leaves, needs)
                while len(self._captured) > self.max_plans:
                    self._captured.pop(next(iter(self._captured)))
            uT = _NodeCapturedFn.apply(u, solve, *leaves)
            return (uT.permute(2, 1, 0) if self.save_every else uT.T), st
        uT = _NodeGenericFn.apply(u, self, ps, st, *leaves)
        return (uT.permute(2, 1, 0) if self.save_every else uT.T), st      # saveat: (D x N x T), the reference's array of the solution

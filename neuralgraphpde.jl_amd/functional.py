"""Differentiable wrappers over the C ABI (the Python stand-in for the `ChainRulesCore.rrule`s the
Julia shim of INTEGRATION.md defines).  All tensors here are in KERNEL layout: features [N][D]
row-major float32 on the GPU (= a Julia (D x N) matrix), weights [in][out] (= Julia (out x in)).
"""
from __future__ import annotations

import os

import torch

from . import _lib


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.ArgumentError(
                _lib.ERR_INVALID_ARGUMENT,
                "the message-passing hot path runs on the MI355X only: move inputs, parameters and "
                "the state to the GPU (there is no CPU fallback)")


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


class _GCNConvFn(torch.autograd.Function):
    """y = act(W (x C (A+I) C) + b)  -- /root/reference/src/layers.jl:200-239.  `edge_weight`: the call's edge_weight argument
    when a gradient is wanted for it (the handle already carries these weights; the tensor enters only as an autograd input)."""

    @staticmethod
    def forward(ctx, x, wt, bias, handle, act, edge_weight=None):
        lib = _lib.load()
        _need_cuda(x, wt, bias)
        n, din = x.shape
        dout = wt.shape[1]
        if wt.shape[0] != din:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"DimensionMismatch: weight is ({dout} x {wt.shape[0]}), x has {din} features")
        x, wt = x.contiguous(), wt.contiguous()
        y = torch.empty((n, dout), dtype=torch.float32, device=x.device)
        need_grad = any(ctx.needs_input_grad)
        agg = torch.empty((n, din), dtype=torch.float32, device=x.device) if (need_grad and dout >= din) else None
        z = torch.empty((n, dout), dtype=torch.float32, device=x.device) if need_grad else None
        ws = _ws(lib.ngpde_gcn_workspace_bytes(handle.ptr, din, dout, 0), x.device)
        _lib.check(lib.ngpde_gcn_forward(handle.ptr, din, dout, act, _lib.ptr(x), _lib.ptr(wt), _lib.ptr(bias),
                                         _lib.ptr(y), _lib.ptr(agg), _lib.ptr(z), _lib.ptr(ws), ws.numel(),
                                         _lib.current_stream()))
        ctx.handle, ctx.act, ctx.dims = handle, act, (n, din, dout)
        ctx.has_bias = bias is not None
        ctx.n_edges = None if edge_weight is None else int(edge_weight.numel())
        ctx.save_for_backward(x, wt, z, agg, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, wt, z, agg, bias = ctx.saved_tensors
        n, din, dout = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dwt = torch.empty_like(wt)
        db = torch.empty((dout,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
        if ctx.n_edges is not None and ctx.needs_input_grad[5]:
            dew = torch.empty((ctx.n_edges,), dtype=torch.float32, device=x.device)
            ws = _ws(lib.ngpde_gcn_backward_ew_workspace_bytes(ctx.handle.ptr, din, dout), x.device)
            _lib.check(lib.ngpde_gcn_backward_ew(ctx.handle.ptr, din, dout, ctx.act, _lib.ptr(x), _lib.ptr(wt), _lib.ptr(bias), _lib.ptr(z),
                                                 _lib.ptr(agg), _lib.ptr(dy), _lib.ptr(dx), _lib.ptr(dwt), _lib.ptr(db), _lib.ptr(dew),
                                                 _lib.ptr(ws), ws.numel(), _lib.current_stream()))
            return dx, dwt, db, None, None, dew
        ws = _ws(lib.ngpde_gcn_workspace_bytes(ctx.handle.ptr, din, dout, 1), x.device)
        _lib.check(lib.ngpde_gcn_backward(ctx.handle.ptr, din, dout, ctx.act, _lib.ptr(x), _lib.ptr(wt), _lib.ptr(z),
                                          _lib.ptr(agg), _lib.ptr(dy), _lib.ptr(dx), _lib.ptr(dwt), _lib.ptr(db),
                                          _lib.ptr(ws), ws.numel(), _lib.current_stream()))
        return dx, dwt, db, None, None, None


def gcn_conv(x, wt, bias, handle, act, edge_weight=None):
    """edge_weight: pass the call's edge_weight tensor when it requires a gradient (src/layers.jl:206-231)"""
    return _GCNConvFn.apply(x, wt, bias, handle, act, edge_weight)


def propagate_copy_xj(x, handle, aggr="+", edge_weight=None, by_source=False):
    """out[i] = aggr_{e: t_e = i} w_e x[s_e]   (no autograd; forward primitive)."""
    lib = _lib.load()
    _need_cuda(x, edge_weight)
    x = x.contiguous()
    out = torch.empty_like(x)
    _lib.check(lib.ngpde_propagate_copy_xj(handle.ptr, x.shape[1], _lib.AGGR[aggr], int(by_source), _lib.ptr(x),
                                           _lib.ptr(edge_weight), _lib.ptr(out), _lib.current_stream()))
    return out


# ---- message-passing primitives (include/ngpde.h, "Message-passing primitives") ---------------------------


def _ptr_array(tensors):
    import ctypes as C
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _int_array(vals):
    import ctypes as C
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])


class _DenseFn(torch.autograd.Function):
    """y = act([X1 | X2 | ...] Wt + b): Lux Dense on a virtual vcat (no concatenation temporary)."""

    @staticmethod
    def forward(ctx, wt, bias, act, row_divs, n, *blocks):
        lib = _lib.load()
        _need_cuda(wt, bias, *blocks)
        blocks = [b.contiguous() for b in blocks]
        wt = wt.contiguous()
        widths = [b.shape[1] for b in blocks]
        din, dout = sum(widths), wt.shape[1]
        if wt.shape[0] != din:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"DimensionMismatch: Dense expects {wt.shape[0]} input features, got {din}")
        for b, rd in zip(blocks, row_divs):
            if b.shape[0] * rd != n and not (rd > 1 and b.shape[0] * rd >= n):
                raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                             f"DimensionMismatch: block with {b.shape[0]} rows (x{rd}) does not cover {n} rows")
        dev = wt.device
        y = torch.empty((n, dout), dtype=torch.float32, device=dev)
        need = any(ctx.needs_input_grad)
        z = torch.empty_like(y) if (need and act != 0) else None
        _lib.check(lib.ngpde_dense_forward(n, len(blocks), _ptr_array(blocks), _int_array(widths), _int_array(row_divs),
                                           dout, act, _lib.ptr(wt), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(z),
                                           _lib.current_stream()))
        ctx.meta = (act, tuple(row_divs), n, widths, dout, bias is not None)
        ctx.save_for_backward(wt, z, *blocks)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        act, row_divs, n, widths, dout, has_bias = ctx.meta
        wt, z, *blocks = ctx.saved_tensors
        dy = dy.contiguous()
        dev = wt.device
        dwt = torch.empty_like(wt)
        db = torch.empty((dout,), dtype=torch.float32, device=dev) if has_bias else None
        dblocks = [torch.empty_like(b) if (ctx.needs_input_grad[5 + i] and row_divs[i] == 1) else None
                   for i, b in enumerate(blocks)]
        ws = _ws(lib.ngpde_dense_workspace_bytes(n, sum(widths), dout), dev)
        _lib.check(lib.ngpde_dense_backward(n, len(blocks), _ptr_array(blocks), _int_array(widths), _int_array(row_divs),
                                            dout, act, _lib.ptr(wt), _lib.ptr(z), _lib.ptr(dy), _ptr_array(dblocks),
                                            _lib.ptr(dwt), _lib.ptr(db), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
        return (dwt, db, None, None, None, *dblocks)


def dense(blocks, wt, bias, act, row_divs=None, n=None):
    """blocks: list of [n_i][w_i] tensors; wt [sum w][dout]; returns [n][dout]."""
    row_divs = list(row_divs) if row_divs is not None else [1] * len(blocks)
    if n is None:
        n = next(b.shape[0] for b, rd in zip(blocks, row_divs) if rd == 1)
    return _DenseFn.apply(wt, bias, act, tuple(row_divs), int(n), *blocks)


class _EdgePermuteFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, handle, inverse):
        lib = _lib.load()
        _need_cuda(x)
        x = x.contiguous()
        out = torch.empty_like(x)
        _lib.check(lib.ngpde_edge_permute(handle.ptr, x.shape[1], int(inverse), _lib.ptr(x), _lib.ptr(out),
                                          _lib.current_stream()))
        ctx.handle, ctx.inverse = handle, inverse
        return out

    @staticmethod
    def backward(ctx, g):
        return _EdgePermuteFn.apply(g.contiguous(), ctx.handle, not ctx.inverse), None, None


def edge_permute(x, handle, inverse=False):
    """[E][d] COO order -> p order (CSR by target); inverse=True for the way back."""
    return _EdgePermuteFn.apply(x, handle, inverse)


class _GatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wx, a, handle, heads, c, slope, n_edges):
        lib = _lib.load()
        _need_cuda(wx, a)
        wx, a = wx.contiguous(), a.contiguous()
        n = wx.shape[0]
        dev = wx.device
        out = torch.empty_like(wx)
        alpha = torch.empty((max(n_edges, 1), heads), dtype=torch.float32, device=dev)
        al = torch.empty((n, heads), dtype=torch.float32, device=dev)
        ar = torch.empty((n, heads), dtype=torch.float32, device=dev)
        _lib.check(lib.ngpde_gat_forward(handle.ptr, heads, c, slope, _lib.ptr(wx), _lib.ptr(a), _lib.ptr(out), _lib.ptr(alpha),
                                         _lib.ptr(al), _lib.ptr(ar), _lib.current_stream()))
        ctx.handle, ctx.meta = handle, (heads, c, slope)
        ctx.save_for_backward(wx, a, al, ar, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        wx, a, al, ar, alpha = ctx.saved_tensors
        heads, c, slope = ctx.meta
        dout = dout.contiguous()
        dwx, da = torch.empty_like(wx), torch.empty_like(a)
        ws = _ws(lib.ngpde_gat_workspace_bytes(ctx.handle.ptr, heads), wx.device)
        _lib.check(lib.ngpde_gat_backward(ctx.handle.ptr, heads, c, slope, _lib.ptr(wx), _lib.ptr(a), _lib.ptr(al), _lib.ptr(ar),
                                          _lib.ptr(alpha), _lib.ptr(dout), _lib.ptr(dwx), _lib.ptr(da), _lib.ptr(ws),
                                          ws.numel(), _lib.current_stream()))
        return dwx, da, None, None, None, None, None


def gat_aggregate(wx, a, handle, heads, c, slope, n_edges):
    return _GatFn.apply(wx, a, handle, heads, c, float(slope), n_edges)


class _GatLayerFn(torch.autograd.Function):
    """The whole GAT-style layer in one launch (ngpde_gat_layer_forward), pullback in two launches + one reduction."""

    @staticmethod
    def forward(ctx, x, wt, a, bias, handle, heads, c, slope, act, n_edges):
        lib = _lib.load()
        _need_cuda(x, wt, a, bias)
        x, wt, a = x.contiguous(), wt.contiguous(), a.contiguous()
        n, dev = x.shape[0], x.device
        need = any(ctx.needs_input_grad)
        y = torch.empty((n, heads * c), dtype=torch.float32, device=dev)
        alpha = torch.empty((max(n_edges, 1), heads), dtype=torch.float32, device=dev) if need else None
        z = torch.empty_like(y) if (need and act not in (0, 1)) else None       # identity / relu: y is enough
        _lib.check(lib.ngpde_gat_layer_forward(handle.ptr, x.shape[1], heads, c, slope, act, _lib.ptr(x), _lib.ptr(wt), _lib.ptr(a),
                                               _lib.ptr(bias), _lib.ptr(y), _lib.ptr(alpha), _lib.ptr(z), _lib.current_stream()))
        ctx.handle, ctx.meta = handle, (heads, c, slope, act, bias is not None)
        ctx.save_for_backward(x, wt, a, alpha, z if z is not None else y)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, wt, a, alpha, yz = ctx.saved_tensors
        heads, c, slope, act, has_bias = ctx.meta
        dy = dy.contiguous()
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dwt, da = torch.empty_like(wt), torch.empty_like(a)
        db = torch.empty((heads * c,), dtype=torch.float32, device=dev) if has_bias else None
        ws = _ws(lib.ngpde_gat_layer_workspace_bytes(ctx.handle.ptr, heads, c), dev)
        _lib.check(lib.ngpde_gat_layer_backward(ctx.handle.ptr, x.shape[1], heads, c, slope, act, _lib.ptr(x), _lib.ptr(wt),
                                                _lib.ptr(a), _lib.ptr(yz), _lib.ptr(alpha), _lib.ptr(dy), _lib.ptr(dx),
                                                _lib.ptr(dwt), _lib.ptr(da), _lib.ptr(db), _lib.ptr(ws), ws.numel(),
                                                _lib.current_stream()))
        return dx, dwt, da, db, None, None, None, None, None, None


def gat_layer_supported(handle, din, heads, c):
    return bool(_lib.load().ngpde_gat_layer_supported(handle.ptr, int(din), int(heads), int(c)))


def gat_layer(x, wt, a, bias, handle, heads, c, slope, act, n_edges):
    """x [N][64], wt [64][heads*c], a (2c x heads) column-major as [heads][2c], bias [heads*c] or None -> y [N][heads*c]"""
    return _GatLayerFn.apply(x, wt, a, bias, handle, int(heads), int(c), float(slope), int(act), int(n_edges))


class _BiasActFn(torch.autograd.Function):
    """y = act(a + addend + b) (ngpde_bias_act_forward): the tail of a layer whose linear part was computed elsewhere."""

    @staticmethod
    def forward(ctx, a, addend, bias, act):
        lib = _lib.load()
        _need_cuda(a, addend, bias)
        a = a.contiguous()
        addend = None if addend is None else addend.contiguous()
        n, d = a.shape
        need = any(ctx.needs_input_grad)
        y = torch.empty_like(a)
        z = torch.empty_like(a) if (need and act not in (0, 1)) else None
        _lib.check(lib.ngpde_bias_act_forward(n, d, act, _lib.ptr(a), _lib.ptr(addend), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(z),
                                              _lib.current_stream()))
        ctx.meta = (act, addend is not None, bias is not None)
        ctx.save_for_backward(z if z is not None else (y if act == 1 else None))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (yz,) = ctx.saved_tensors
        act, has_add, has_bias = ctx.meta
        dy = dy.contiguous()
        n, d = dy.shape
        if act == 0 and not has_bias:
            return dy, (dy if has_add else None), None, None
        dz = torch.empty_like(dy) if act != 0 else dy
        db = torch.empty((d,), dtype=torch.float32, device=dy.device) if has_bias else None
        ws = _ws(lib.ngpde_bias_act_workspace_bytes(d), dy.device) if has_bias else None
        # identity: dz aliases dy, the library skips the element-wise pass and only sums the columns
        _lib.check(lib.ngpde_bias_act_backward(n, d, act, _lib.ptr(dy), _lib.ptr(yz), _lib.ptr(dz), _lib.ptr(db), _lib.ptr(ws),
                                               ws.numel() if ws is not None else 0, _lib.current_stream()))
        return dz, (dz if has_add else None), db, None


def bias_act(a, addend, bias, act):
    return _BiasActFn.apply(a, addend, bias, int(act))


class _PropagateFn(torch.autograd.Function):
    """propagate(e_mul_xj / copy_xj, g, +) with its pullback (the transposed aggregation)."""

    @staticmethod
    def forward(ctx, x, handle, edge_weight):
        ctx.handle = handle
        ctx.save_for_backward(edge_weight)
        return propagate_copy_xj(x, handle, "+", edge_weight, by_source=False)

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        return propagate_copy_xj(dy.contiguous(), ctx.handle, "+", w, by_source=True), None, None


def propagate_sum(x, handle, edge_weight=None):
    return _PropagateFn.apply(x, handle, edge_weight)


def spectral_weights(e, n):
    lib = _lib.load()
    _need_cuda(e)
    e = e.contiguous().reshape(-1)
    w = torch.empty_like(e)
    _lib.check(lib.ngpde_spectral_weights(e.numel(), int(n), _lib.ptr(e), _lib.ptr(w), _lib.current_stream()))
    return w


def _mlp_struct(wts, bs, acts):
    m = _lib.Mlp()
    m.n_layers = len(wts)
    for l, w in enumerate(wts):
        m.dims[l], m.dims[l + 1] = w.shape[0], w.shape[1]
        m.act[l] = acts[l]
        m.weight[l] = w.data_ptr()
        m.bias[l] = bs[l].data_ptr() if bs[l] is not None else None
    return m


class _EdgeLayerFn(torch.autograd.Function):
    """ExplicitEdgeConv / VMHConv / MPPDEConv in ONE library call (ngpde_edge_layer_forward), the pullback in one
    (ngpde_edge_layer_backward): the split of phi's first weight, P / Q / E, the message path, the node update and every saved
    activation live behind the C ABI, in one workspace.  tensors = state blocks, then (weight, bias) of every phi layer, then of
    every update layer (bias None where the layer has none)."""

    @staticmethod
    def forward(ctx, handle, kind, aggr, consts, phi_acts, upd_acts, grad_on, n_state, *tensors):
        import ctypes as C
        lib = _lib.load()
        _need_cuda(*[t for t in tensors if t is not None], *[t for t in consts.values() if t is not None])
        n_phi, n_upd = len(phi_acts), len(upd_acts)
        tensors = [None if t is None else t.contiguous() for t in tensors]
        state = tensors[:n_state]
        pw = tensors[n_state:n_state + 2 * n_phi]
        uw = tensors[n_state + 2 * n_phi:]
        for l in range(1, n_phi):
            if pw[2 * l].shape[0] != pw[2 * l - 2].shape[1]:
                raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, f"DimensionMismatch: Dense expects {pw[2 * l].shape[0]} input "
                                             f"features, got {pw[2 * l - 2].shape[1]}")
        for l in range(1, n_upd):
            if uw[2 * l].shape[0] != uw[2 * l - 2].shape[1]:
                raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, f"DimensionMismatch: Dense expects {uw[2 * l].shape[0]} input "
                                             f"features, got {uw[2 * l - 2].shape[1]}")
        d = _lib.EdgeLayer()
        d.kind, d.aggr, d.n_state = kind, aggr, n_state
        for k, t in enumerate(state):
            d.state[k], d.state_width[k] = t.data_ptr(), t.shape[1]
        keep = []
        for name in ("node_feat", "pos", "edge_feat", "theta"):
            t = consts.get(name)
            if t is not None and t.shape[1] > 0:
                t = t.contiguous()
                keep.append(t)
                setattr(d, name, t.data_ptr())
                setattr(d, name + "_width", t.shape[1])
        d.phi = _mlp_struct(pw[0::2], pw[1::2], phi_acts)
        if n_upd:
            d.update = _mlp_struct(uw[0::2], uw[1::2], upd_acts)
        # (grad_on: torch.is_grad_enabled() of the CALLER -- under no_grad() a leaf parameter still reports requires_grad, and nothing
        # is to be kept for a pullback that cannot come)
        training = grad_on and any(ctx.needs_input_grad)
        dev = state[0].device
        n_nodes = state[0].shape[0]
        out_w = (uw[-2] if n_upd else pw[-2]).shape[1]
        nbytes = int(lib.ngpde_edge_layer_workspace_bytes(handle.ptr, C.byref(d), int(training)))
        ws = _ws(nbytes, dev)
        y = torch.empty((n_nodes, out_w), dtype=torch.float32, device=dev)
        _lib.check(lib.ngpde_edge_layer_forward(handle.ptr, C.byref(d), int(training), _lib.ptr(y), _lib.ptr(ws), ws.numel(),
                                                _lib.current_stream()))
        if training:
            ctx.desc, ctx.ws, ctx.keep, ctx.handle = d, ws, keep, handle
            ctx.counts = (n_state, n_phi, n_upd)
            ctx.present = [t is not None for t in tensors]
            ctx.save_for_backward(*[t for t in tensors if t is not None])
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        lib = _lib.load()
        n_state, n_phi, n_upd = ctx.counts
        it = iter(ctx.saved_tensors)
        tensors = [next(it) if pr else None for pr in ctx.present]
        dev = dy.device
        dy = dy.contiguous()
        grads = [None] * len(tensors)
        dstate = (C.c_void_p * 4)()
        for k in range(n_state):
            if ctx.needs_input_grad[8 + k]:
                grads[k] = torch.empty_like(tensors[k])
                dstate[k] = grads[k].data_ptr()
        gphi, gupd = _lib.MlpGrad(), _lib.MlpGrad()
        for base, n, gs in ((n_state, n_phi, gphi), (n_state + 2 * n_phi, n_upd, gupd)):
            for l in range(n):
                grads[base + 2 * l] = torch.empty_like(tensors[base + 2 * l])
                gs.dweight[l] = grads[base + 2 * l].data_ptr()
                if tensors[base + 2 * l + 1] is not None:
                    grads[base + 2 * l + 1] = torch.empty_like(tensors[base + 2 * l + 1])
                    gs.dbias[l] = grads[base + 2 * l + 1].data_ptr()
        _lib.check(lib.ngpde_edge_layer_backward(ctx.handle.ptr, C.byref(ctx.desc), _lib.ptr(dy), dstate, C.byref(gphi), C.byref(gupd),
                                                 _lib.ptr(ctx.ws), ctx.ws.numel(), _lib.current_stream()))
        # (the workspace holds every saved activation and is only read by the pullback: it stays with the node, so a second backward
        # through a retained graph finds it, and goes when autograd frees the node)
        return (None, None, None, None, None, None, None, None, *grads)


def edge_layer(handle, kind, aggr, state, phi, update=(), node_feat=None, pos=None, edge_feat=None, theta=None):
    """state: list of [N][w] blocks; phi / update: lists of (weight [in][out], bias or None, activation code).  Returns [N][out]."""
    tensors = list(state)
    for wt, b, _ in list(phi) + list(update):
        tensors += [wt, b]
    consts = {"node_feat": node_feat, "pos": pos, "edge_feat": edge_feat, "theta": theta}
    return _EdgeLayerFn.apply(handle, int(kind), _lib.AGGR[aggr] if isinstance(aggr, str) else int(aggr), consts,
                              tuple(a for _, _, a in phi), tuple(a for _, _, a in update), torch.is_grad_enabled(), len(state), *tensors)


class _GnoLayerFn(torch.autograd.Function):
    """GNOConv in ONE library call (ngpde_gno_layer_forward), its pullback in one (ngpde_gno_layer_backward).
    tensors = h, linear weight, linear bias (or None), then (weight, bias) of every phi layer."""

    @staticmethod
    def forward(ctx, handle, cin, cout, aggr, act, consts, phi_acts, grad_on, *tensors):
        import ctypes as C
        lib = _lib.load()
        _need_cuda(*[t for t in tensors if t is not None], *[t for t in consts.values() if t is not None])
        tensors = [None if t is None else t.contiguous() for t in tensors]
        h, lwt, lb = tensors[:3]
        pw = tensors[3:]
        n_phi = len(phi_acts)
        for l in range(1, n_phi):
            if pw[2 * l].shape[0] != pw[2 * l - 2].shape[1]:
                raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, f"DimensionMismatch: Dense expects {pw[2 * l].shape[0]} input "
                                             f"features, got {pw[2 * l - 2].shape[1]}")
        if tuple(lwt.shape) != (cin, cout) or h.shape[1] != cin:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, f"DimensionMismatch: Dense expects {lwt.shape[0]} input features, got {h.shape[1]}")
        d = _lib.GnoLayer()
        d.in_chs, d.out_chs, d.aggr, d.act = cin, cout, aggr, act
        d.h, d.weight, d.bias = h.data_ptr(), lwt.data_ptr(), (lb.data_ptr() if lb is not None else None)
        keep = []
        for name in ("node_feat", "edge_feat"):
            t = consts.get(name)
            if t is not None and t.shape[1] > 0:
                t = t.contiguous()
                keep.append(t)
                setattr(d, name, t.data_ptr())
                setattr(d, name + "_width", t.shape[1])
        d.phi = _mlp_struct(pw[0::2], pw[1::2], phi_acts)
        # (grad_on: torch.is_grad_enabled() of the CALLER -- under no_grad() a leaf parameter still reports requires_grad, and nothing
        # is to be kept for a pullback that cannot come)
        training = grad_on and any(ctx.needs_input_grad)
        dev = h.device
        ws = _ws(int(lib.ngpde_gno_layer_workspace_bytes(handle.ptr, C.byref(d), int(training))), dev)
        y = torch.empty((h.shape[0], cout), dtype=torch.float32, device=dev)
        _lib.check(lib.ngpde_gno_layer_forward(handle.ptr, C.byref(d), int(training), _lib.ptr(y), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
        if training:
            ctx.desc, ctx.ws, ctx.keep, ctx.handle, ctx.n_phi = d, ws, keep, handle, n_phi
            ctx.present = [t is not None for t in tensors]
            ctx.save_for_backward(*[t for t in tensors if t is not None])
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        lib = _lib.load()
        it = iter(ctx.saved_tensors)
        tensors = [next(it) if pr else None for pr in ctx.present]
        grads = [None] * len(tensors)
        dy = dy.contiguous()
        if ctx.needs_input_grad[8]:
            grads[0] = torch.empty_like(tensors[0])
        grads[1] = torch.empty_like(tensors[1])
        if tensors[2] is not None:
            grads[2] = torch.empty_like(tensors[2])
        gphi = _lib.MlpGrad()
        for l in range(ctx.n_phi):
            grads[3 + 2 * l] = torch.empty_like(tensors[3 + 2 * l])
            gphi.dweight[l] = grads[3 + 2 * l].data_ptr()
            if tensors[4 + 2 * l] is not None:
                grads[4 + 2 * l] = torch.empty_like(tensors[4 + 2 * l])
                gphi.dbias[l] = grads[4 + 2 * l].data_ptr()
        _lib.check(lib.ngpde_gno_layer_backward(ctx.handle.ptr, C.byref(ctx.desc), _lib.ptr(dy), _lib.ptr(grads[0]), C.byref(gphi), _lib.ptr(grads[1]),
                                                _lib.ptr(grads[2]), _lib.ptr(ctx.ws), ctx.ws.numel(), _lib.current_stream()))
        return (None, None, None, None, None, None, None, None, *grads)


def gno_layer(handle, cin, cout, aggr, act, h, lwt, lb, phi, node_feat=None, edge_feat=None):
    """phi: list of (weight [in][out], bias or None, activation code); lwt [in][out], lb [out] or None.  Returns [N][out]."""
    tensors = [h, lwt, lb]
    for wt, b, _ in phi:
        tensors += [wt, b]
    return _GnoLayerFn.apply(handle, int(cin), int(cout), _lib.AGGR[aggr] if isinstance(aggr, str) else int(aggr), int(act),
                             {"node_feat": node_feat, "edge_feat": edge_feat}, tuple(a for _, _, a in phi), torch.is_grad_enabled(), *tensors)


# ---- weight-sized rearrangements as library launches (row_blocks.hip) ---------------------------------------------------------

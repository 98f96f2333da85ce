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


def _check_blocks(blocks, row_divs, n, wt):
    widths = [b.shape[1] for b in blocks]
    if wt.shape[0] != sum(widths):
        raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                     f"DimensionMismatch: Dense expects {wt.shape[0]} input features, got {sum(widths)}")
    for b, rd in zip(blocks, row_divs):
        if b.shape[0] * rd != n and not (rd > 1 and b.shape[0] * rd >= n):
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"DimensionMismatch: block with {b.shape[0]} rows (x{rd}) does not cover {n} rows")
    return widths


def _dense_backward_call(lib, n, blocks, widths, row_divs, dout, act, wt, z, dy, want, has_bias):
    """one ngpde_dense_backward: returns (dwt, db, [dblock or None])"""
    dev = wt.device
    dwt = torch.empty_like(wt)
    db = torch.empty((dout,), dtype=torch.float32, device=dev) if has_bias else None
    dblocks = [torch.empty_like(b) if (w and rd == 1) else None for b, w, rd in zip(blocks, want, row_divs)]
    ws = _ws(lib.ngpde_dense_workspace_bytes(n, sum(widths), dout), dev)
    _lib.check(lib.ngpde_dense_backward(n, len(blocks), _ptr_array(blocks), _int_array(widths), _int_array(row_divs),
                                        dout, act, _lib.ptr(wt), _lib.ptr(z), _lib.ptr(dy), _ptr_array(dblocks),
                                        _lib.ptr(dwt), _lib.ptr(db), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
    return dwt, db, dblocks


class _DensePairFn(torch.autograd.Function):
    """(ya, yb[, x]) = (Dense_a(blocks_a), Dense_b(blocks_b)[, blocks_a[0]]): ngpde_dense_pair_forward -- one pass over a shared
    leading 64-wide block (the node-level target / source halves of a message MLP's first layer), two launches otherwise.
    passthrough: the shared block comes back as a third output; a consumer that reads it through that output (the node update
    psi of MPPDEConv) delivers its gradient HERE, where ngpde_dense_pair_backward adds it while it writes the block's gradient --
    no separate accumulation pass over the [N][64] arrays."""

    @staticmethod
    def forward(ctx, wta, ba, acta, rda, wtb, bb, actb, rdb, n, na, passthrough, *blocks):
        lib = _lib.load()
        _need_cuda(wta, ba, wtb, bb, *blocks)
        x_in = blocks[0]
        blocks = [b.contiguous() for b in blocks]
        wta, wtb = wta.contiguous(), wtb.contiguous()
        A, B = blocks[:na], blocks[na:]
        wa, wb = _check_blocks(A, rda, n, wta), _check_blocks(B, rdb, n, wtb)
        dev = wta.device
        ya = torch.empty((n, wta.shape[1]), dtype=torch.float32, device=dev)
        yb = torch.empty((n, wtb.shape[1]), dtype=torch.float32, device=dev)
        need = any(ctx.needs_input_grad)
        za = torch.empty_like(ya) if (need and acta != 0) else None
        zb = torch.empty_like(yb) if (need and actb != 0) else None
        _lib.check(lib.ngpde_dense_pair_forward(
            n, len(A), _ptr_array(A), _int_array(wa), _int_array(rda), wta.shape[1], acta, _lib.ptr(wta), _lib.ptr(ba), _lib.ptr(ya),
            _lib.ptr(za), len(B), _ptr_array(B), _int_array(wb), _int_array(rdb), wtb.shape[1], actb, _lib.ptr(wtb), _lib.ptr(bb),
            _lib.ptr(yb), _lib.ptr(zb), _lib.current_stream()))
        ctx.meta = (acta, tuple(rda), actb, tuple(rdb), n, na, wa, wb, ba is not None, bb is not None, passthrough)
        ctx.set_materialize_grads(False)   # an unused pass-through output must not cost a zero array and an addend read
        ctx.save_for_backward(wta, wtb, za, zb, *blocks)
        if passthrough:
            return ya, yb, x_in.view_as(x_in)
        return ya, yb

    @staticmethod
    def backward(ctx, dya, dyb, dxp=None):
        lib = _lib.load()
        acta, rda, actb, rdb, n, na, wa, wb, has_ba, has_bb, passthrough = ctx.meta
        wta, wtb, za, zb, *blocks = ctx.saved_tensors
        A, B = blocks[:na], blocks[na:]
        want = ctx.needs_input_grad[11:]
        dev = wta.device
        if dya is None:
            dya = torch.zeros((n, wta.shape[1]), dtype=torch.float32, device=dev)
        if dyb is None:
            dyb = torch.zeros((n, wtb.shape[1]), dtype=torch.float32, device=dev)
        dya, dyb = dya.contiguous(), dyb.contiguous()
        if dxp is not None:
            dxp = dxp.contiguous()
        # one launch for both pullbacks when the pair shares its 64-wide leading block and only that block wants a gradient:
        # dx arrives already summed, the pass-through consumer's gradient included (ngpde_dense_pair_backward)
        shared = (acta == 0 and actb == 0 and wta.shape[1] == 64 and wtb.shape[1] == 64 and A[0].data_ptr() == B[0].data_ptr()
                  and (want[0] or want[na]) and not any(w and rd == 1 for w, rd in zip(want[1:na], rda[1:]))
                  and not any(w and rd == 1 for w, rd in zip(want[na + 1:], rdb[1:])))
        if shared:
            wsb = int(lib.ngpde_dense_pair_backward_workspace_bytes(n, len(A), _ptr_array(A), _int_array(wa), _int_array(rda), len(B),
                                                                    _ptr_array(B), _int_array(wb), _int_array(rdb), 64))
            if wsb > 0:
                dwta, dwtb = torch.empty_like(wta), torch.empty_like(wtb)
                dba = torch.empty((64,), dtype=torch.float32, device=dev) if has_ba else None
                dbb = torch.empty((64,), dtype=torch.float32, device=dev) if has_bb else None
                dx = torch.empty_like(A[0])
                ws = _ws(wsb, dev)
                _lib.check(lib.ngpde_dense_pair_backward(
                    n, len(A), _ptr_array(A), _int_array(wa), _int_array(rda), _lib.ptr(wta), _lib.ptr(dya), _lib.ptr(dwta), _lib.ptr(dba),
                    len(B), _ptr_array(B), _int_array(wb), _int_array(rdb), _lib.ptr(wtb), _lib.ptr(dyb), _lib.ptr(dwtb), _lib.ptr(dbb),
                    64, _lib.ptr(dx), _lib.ptr(dxp), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
                dA, dB = [None] * len(A), [None] * len(B)
                if want[0]:
                    dA[0] = dx
                else:
                    dB[0] = dx
                return (dwta, dba, None, None, dwtb, dbb, None, None, None, None, None, *dA, *dB)
        dwta, dba, dA = _dense_backward_call(lib, n, A, wa, rda, wta.shape[1], acta, wta, za, dya, want[:na], has_ba)
        dwtb, dbb, dB = _dense_backward_call(lib, n, B, wb, rdb, wtb.shape[1], actb, wtb, zb, dyb, want[na:], has_bb)
        if dxp is not None and want[0]:
            if dA[0] is None:
                dA[0] = dxp
            else:                    # one library launch (the Runge-Kutta combination kernel: out = 1 * base + 1 * term), no torch add
                import ctypes as C
                out = torch.empty_like(dA[0])
                _lib.check(lib.ngpde_rk_stage_combine(out.numel(), 1.0, _lib.ptr(dA[0].contiguous()), 1, (C.c_void_p * 1)(dxp.data_ptr()),
                                                      (C.c_float * 1)(1.0), _lib.ptr(out), _lib.current_stream()))
                dA[0] = out
        return (dwta, dba, None, None, dwtb, dbb, None, None, None, None, None, *dA, *dB)


def dense_pair(blocks_a, wta, ba, acta, blocks_b, wtb, bb, actb, row_divs_a=None, row_divs_b=None, n=None, passthrough=False):
    """Two Dense layers whose block lists start with the same tensor; returns (ya, yb), with passthrough=True (ya, yb, x) where x
    is blocks_a[0] routed through this node (use it for further consumers of the block: their gradient is then folded into the
    block's gradient by the pair's own pullback launch)."""
    rda = list(row_divs_a) if row_divs_a is not None else [1] * len(blocks_a)
    rdb = list(row_divs_b) if row_divs_b is not None else [1] * len(blocks_b)
    if n is None:
        n = next(b.shape[0] for b, rd in zip(blocks_a, rda) if rd == 1)
    return _DensePairFn.apply(wta, ba, acta, tuple(rda), wtb, bb, actb, tuple(rdb), int(n), len(blocks_a), bool(passthrough),
                              *blocks_a, *blocks_b)


class _DenseChain2Fn(torch.autograd.Function):
    """y = act2(act1([X1 | ...] W1t + b1) W2t + b2): ngpde_dense_chain2_forward -- the intermediate stays on chip when the shape
    allows (inference: nothing but y is written; training keeps z1 / a1 for the two pullbacks)."""

    @staticmethod
    def forward(ctx, wt1, b1, act1, wt2, b2, act2, row_divs, n, *blocks):
        lib = _lib.load()
        _need_cuda(wt1, b1, wt2, b2, *blocks)
        blocks = [b.contiguous() for b in blocks]
        wt1, wt2 = wt1.contiguous(), wt2.contiguous()
        widths = _check_blocks(blocks, row_divs, n, wt1)
        dmid, dout = wt1.shape[1], wt2.shape[1]
        if wt2.shape[0] != dmid:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"DimensionMismatch: second Dense expects {wt2.shape[0]} input features, got {dmid}")
        dev = wt1.device
        need = any(ctx.needs_input_grad)
        fused = bool(lib.ngpde_dense_chain2_fused(n, len(blocks), _ptr_array(blocks), _int_array(widths), _int_array(row_divs), dmid, dout))
        y = torch.empty((n, dout), dtype=torch.float32, device=dev)
        a1 = torch.empty((n, dmid), dtype=torch.float32, device=dev) if (need or not fused) else None
        z1 = torch.empty((n, dmid), dtype=torch.float32, device=dev) if (need and act1 != 0) else None
        z2 = torch.empty_like(y) if (need and act2 != 0) else None
        _lib.check(lib.ngpde_dense_chain2_forward(n, len(blocks), _ptr_array(blocks), _int_array(widths), _int_array(row_divs), dmid, act1,
                                                  _lib.ptr(wt1), _lib.ptr(b1), _lib.ptr(a1), _lib.ptr(z1), dout, act2, _lib.ptr(wt2),
                                                  _lib.ptr(b2), _lib.ptr(y), _lib.ptr(z2), _lib.current_stream()))
        ctx.meta = (act1, act2, tuple(row_divs), n, widths, dmid, dout, b1 is not None, b2 is not None)
        ctx.save_for_backward(wt1, wt2, a1, z1, z2, *blocks)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        act1, act2, row_divs, n, widths, dmid, dout, has_b1, has_b2 = ctx.meta
        wt1, wt2, a1, z1, z2, *blocks = ctx.saved_tensors
        dwt2, db2, (da1,) = _dense_backward_call(lib, n, [a1], [dmid], [1], dout, act2, wt2, z2, dy.contiguous(), [True], has_b2)
        dwt1, db1, dblocks = _dense_backward_call(lib, n, blocks, widths, row_divs, dmid, act1, wt1, z1, da1, ctx.needs_input_grad[8:], has_b1)
        return (dwt1, db1, None, dwt2, db2, None, None, None, *dblocks)


def dense_chain2(blocks, wt1, b1, act1, wt2, b2, act2, row_divs=None, n=None):
    """Chain(Dense, Dense) on a virtual vcat of blocks; returns [n][dout2]."""
    row_divs = list(row_divs) if row_divs is not None else [1] * len(blocks)
    if n is None:
        n = next(b.shape[0] for b, rd in zip(blocks, row_divs) if rd == 1)
    return _DenseChain2Fn.apply(wt1, b1, act1, wt2, b2, act2, tuple(row_divs), int(n), *blocks)


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


class _EdgeCombineFn(torch.autograd.Function):
    """a_p = act(P[t_p] + Q[s_p] + E_p): gather at t + gather at s + first Dense layer of the message MLP."""

    @staticmethod
    def forward(ctx, P, Q, Eterm, handle, act, n_edges):
        lib = _lib.load()
        _need_cuda(P, Q, Eterm)
        ref = next(t for t in (P, Q, Eterm) if t is not None)
        h = ref.shape[1]
        P = None if P is None else P.contiguous()
        Q = None if Q is None else Q.contiguous()
        Eterm = None if Eterm is None else Eterm.contiguous()
        a = torch.empty((n_edges, h), dtype=torch.float32, device=ref.device)
        z = torch.empty_like(a) if act != 0 else None
        _lib.check(lib.ngpde_edge_combine_forward(handle.ptr, h, act, _lib.ptr(P), _lib.ptr(Q), _lib.ptr(Eterm), _lib.ptr(a),
                                                  _lib.ptr(z), _lib.current_stream()))
        ctx.handle, ctx.act, ctx.h = handle, act, h
        ctx.shapes = (None if P is None else P.shape, None if Q is None else Q.shape, Eterm is not None)
        ctx.save_for_backward(z)
        return a

    @staticmethod
    def backward(ctx, da):
        lib = _lib.load()
        (z,) = ctx.saved_tensors
        da = da.contiguous()
        ps, qs, has_e = ctx.shapes
        dev = da.device
        dz = torch.empty_like(da)
        dP = torch.empty(ps, dtype=torch.float32, device=dev) if (ps is not None and ctx.needs_input_grad[0]) else None
        dQ = torch.empty(qs, dtype=torch.float32, device=dev) if (qs is not None and ctx.needs_input_grad[1]) else None
        _lib.check(lib.ngpde_edge_combine_backward(ctx.handle.ptr, ctx.h, ctx.act, _lib.ptr(da), _lib.ptr(z), _lib.ptr(dz),
                                                   _lib.ptr(dP), _lib.ptr(dQ), _lib.current_stream()))
        return dP, dQ, (dz if (has_e and ctx.needs_input_grad[2]) else None), None, None, None


def edge_combine(P, Q, Eterm, handle, act, n_edges):
    return _EdgeCombineFn.apply(P, Q, Eterm, handle, act, n_edges)


class _SegmentReduceFn(torch.autograd.Function):
    """aggregate_neighbors(g, aggr, m): segmented reduction over each node's incoming edges."""

    @staticmethod
    def forward(ctx, M, handle, aggr, n_nodes):
        lib = _lib.load()
        _need_cuda(M)
        M = M.contiguous()
        d = M.shape[1]
        out = torch.empty((n_nodes, d), dtype=torch.float32, device=M.device)
        _lib.check(lib.ngpde_segment_reduce_forward(handle.ptr, d, aggr, _lib.ptr(M), _lib.ptr(out), _lib.current_stream()))
        ctx.handle, ctx.aggr, ctx.d = handle, aggr, d
        ctx.save_for_backward(M, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        M, out = ctx.saved_tensors
        dout = dout.contiguous()
        dM = torch.empty_like(M)
        _lib.check(lib.ngpde_segment_reduce_backward(ctx.handle.ptr, ctx.d, ctx.aggr, _lib.ptr(M), _lib.ptr(out),
                                                     _lib.ptr(dout), _lib.ptr(dM), _lib.current_stream()))
        return dM, None, None, None


def segment_reduce(M, handle, aggr, n_nodes):
    return _SegmentReduceFn.apply(M, handle, _lib.AGGR[aggr] if isinstance(aggr, str) else aggr, n_nodes)


class _GnoContractFn(torch.autograd.Function):
    """m_e = reshape(K_e, out, in) * h[:, s_e]   (NNlib.batched_mul, src/layers.jl:527-530)."""

    @staticmethod
    def forward(ctx, K, h, handle, cin, cout):
        lib = _lib.load()
        _need_cuda(K, h)
        K, h = K.contiguous(), h.contiguous()
        m = torch.empty((K.shape[0], cout), dtype=torch.float32, device=K.device)
        _lib.check(lib.ngpde_gno_contract_forward(handle.ptr, cin, cout, _lib.ptr(K), _lib.ptr(h), _lib.ptr(m),
                                                  _lib.current_stream()))
        ctx.handle, ctx.dims = handle, (cin, cout)
        ctx.save_for_backward(K, h)
        return m

    @staticmethod
    def backward(ctx, dm):
        lib = _lib.load()
        K, h = ctx.saved_tensors
        cin, cout = ctx.dims
        dm = dm.contiguous()
        dK = torch.empty_like(K) if ctx.needs_input_grad[0] else None
        dh = torch.empty_like(h) if ctx.needs_input_grad[1] else None
        ws = _ws(K.shape[0] * cin * 4, K.device)
        _lib.check(lib.ngpde_gno_contract_backward(ctx.handle.ptr, cin, cout, _lib.ptr(K), _lib.ptr(h), _lib.ptr(dm),
                                                   _lib.ptr(dK), _lib.ptr(dh), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
        return dK, dh, None, None, None


def gno_contract(K, h, handle, cin, cout):
    return _GnoContractFn.apply(K, h, handle, cin, cout)


class _GnoApplyFn(torch.autograd.Function):
    """Reassociated GNOConv message  m_e = T_{s_e} z_e + Bh_{s_e}  (include/ngpde.h: ngpde_gno_apply_forward)."""

    @staticmethod
    def forward(ctx, T, Bh, z, handle, cout, kdim):
        lib = _lib.load()
        _need_cuda(T, z)
        T, z = T.contiguous(), z.contiguous()
        Bh = Bh.contiguous() if Bh is not None else None
        m = torch.empty((z.shape[0], cout), dtype=torch.float32, device=z.device)
        _lib.check(lib.ngpde_gno_apply_forward(handle.ptr, cout, kdim, _lib.ptr(T), _lib.ptr(Bh), _lib.ptr(z), _lib.ptr(m),
                                               _lib.current_stream()))
        ctx.handle, ctx.dims, ctx.has_bh = handle, (cout, kdim), Bh is not None
        ctx.save_for_backward(T, z)
        return m

    @staticmethod
    def backward(ctx, dm):
        lib = _lib.load()
        T, z = ctx.saved_tensors
        cout, kdim = ctx.dims
        dm = dm.contiguous()
        dT = torch.empty_like(T) if ctx.needs_input_grad[0] else None
        dBh = (torch.empty((T.shape[0], cout), dtype=torch.float32, device=T.device)
               if ctx.has_bh and ctx.needs_input_grad[1] else None)
        dz = torch.empty_like(z) if ctx.needs_input_grad[2] else None
        _lib.check(lib.ngpde_gno_apply_backward(ctx.handle.ptr, cout, kdim, _lib.ptr(T), _lib.ptr(z), _lib.ptr(dm),
                                                _lib.ptr(dT), _lib.ptr(dBh), _lib.ptr(dz), _lib.current_stream()))
        return dT, dBh, dz, None, None, None


def gno_apply_supported(cout, kdim):
    return bool(_lib.load().ngpde_gno_apply_supported(int(cout), int(kdim)))


def gno_apply(T, Bh, z, handle, cout, kdim):
    return _GnoApplyFn.apply(T, Bh, z, handle, cout, kdim)


class _GnoMessageFn(torch.autograd.Function):
    """m_e = T_{s_e} act1(P[t_e] + Q[s_e] + E_e) + Bh_{s_e} in ONE launch (ngpde_gno_message_forward): the per-edge input of the
    reassociated GNOConv message is formed while the per-source GEMM stages its rows.  act1 in {identity, relu}: the activated
    input kept for the pullback also tells act1' (pullback = ngpde_gno_apply_backward + ngpde_edge_combine_backward)."""

    @staticmethod
    def forward(ctx, P, Q, Eterm, T, Bh, handle, act1, cout, kdim, n_edges):
        lib = _lib.load()
        _need_cuda(P, Q, Eterm, T, Bh)
        P = None if P is None else P.contiguous()
        Q = None if Q is None else Q.contiguous()
        Eterm = None if Eterm is None else Eterm.contiguous()
        T = T.contiguous()
        Bh = None if Bh is None else Bh.contiguous()
        dev = T.device
        need = any(ctx.needs_input_grad)
        a = torch.empty((n_edges, kdim), dtype=torch.float32, device=dev) if need else None
        m = torch.empty((n_edges, cout), dtype=torch.float32, device=dev)
        _lib.check(lib.ngpde_gno_message_forward(handle.ptr, cout, kdim, act1, _lib.ptr(P), _lib.ptr(Q), _lib.ptr(Eterm), _lib.ptr(T),
                                                 _lib.ptr(Bh), _lib.ptr(a), _lib.ptr(m), _lib.current_stream()))
        ctx.handle, ctx.meta = handle, (act1, cout, kdim)
        ctx.shapes = (None if P is None else P.shape, None if Q is None else Q.shape, Eterm is not None, Bh is not None)
        ctx.save_for_backward(T, a)
        return m

    @staticmethod
    def backward(ctx, dm):
        lib = _lib.load()
        T, a = ctx.saved_tensors
        act1, cout, kdim = ctx.meta
        pshape, qshape, has_e, has_bh = ctx.shapes
        dm = dm.contiguous()
        dev = dm.device
        stream = _lib.current_stream()
        dT = torch.empty_like(T) if ctx.needs_input_grad[3] else None
        dBh = torch.empty((T.shape[0], cout), dtype=torch.float32, device=dev) if (has_bh and ctx.needs_input_grad[4]) else None
        da = torch.empty_like(a)
        _lib.check(lib.ngpde_gno_apply_backward(ctx.handle.ptr, cout, kdim, _lib.ptr(T), _lib.ptr(a), _lib.ptr(dm), _lib.ptr(dT),
                                                _lib.ptr(dBh), _lib.ptr(da), stream))
        dz = torch.empty_like(a)
        dP = torch.empty(pshape, dtype=torch.float32, device=dev) if (pshape is not None and ctx.needs_input_grad[0]) else None
        dQ = torch.empty(qshape, dtype=torch.float32, device=dev) if (qshape is not None and ctx.needs_input_grad[1]) else None
        _lib.check(lib.ngpde_edge_combine_backward(ctx.handle.ptr, kdim, act1, _lib.ptr(da), _lib.ptr(a), _lib.ptr(dz), _lib.ptr(dP),
                                                   _lib.ptr(dQ), stream))
        return dP, dQ, (dz if (has_e and ctx.needs_input_grad[2]) else None), dT, dBh, None, None, None, None, None


class _GnoMessageAggFn(torch.autograd.Function):
    """agg = aggregate_neighbors(g, aggr, m) of the fused message above, aggr in {+, mean} (src/layers.jl:527-534): forward = the
    message launch + the segmented reduction; the pullback forms dm_e = dagg[t_e] (/ deg) inside the per-source launch
    (ngpde_gno_message_backward_from_nodes) instead of writing and re-reading an [E][out] array."""

    @staticmethod
    def forward(ctx, P, Q, Eterm, T, Bh, handle, act1, cout, kdim, n_edges, aggr, n_nodes):
        lib = _lib.load()
        _need_cuda(P, Q, Eterm, T, Bh)
        P = None if P is None else P.contiguous()
        Q = None if Q is None else Q.contiguous()
        Eterm = None if Eterm is None else Eterm.contiguous()
        T = T.contiguous()
        Bh = None if Bh is None else Bh.contiguous()
        dev = T.device
        stream = _lib.current_stream()
        need = any(ctx.needs_input_grad)
        a = torch.empty((n_edges, kdim), dtype=torch.float32, device=dev) if need else None
        m = torch.empty((n_edges, cout), dtype=torch.float32, device=dev)
        _lib.check(lib.ngpde_gno_message_forward(handle.ptr, cout, kdim, act1, _lib.ptr(P), _lib.ptr(Q), _lib.ptr(Eterm), _lib.ptr(T),
                                                 _lib.ptr(Bh), _lib.ptr(a), _lib.ptr(m), stream))
        agg = torch.empty((n_nodes, cout), dtype=torch.float32, device=dev)
        _lib.check(lib.ngpde_segment_reduce_forward(handle.ptr, cout, aggr, _lib.ptr(m), _lib.ptr(agg), stream))
        ctx.handle, ctx.meta = handle, (act1, cout, kdim, aggr)
        ctx.shapes = (None if P is None else P.shape, None if Q is None else Q.shape, Eterm is not None, Bh is not None)
        ctx.save_for_backward(T, a)
        return agg

    @staticmethod
    def backward(ctx, dagg):
        lib = _lib.load()
        T, a = ctx.saved_tensors
        act1, cout, kdim, aggr = ctx.meta
        pshape, qshape, has_e, has_bh = ctx.shapes
        dev = dagg.device
        if aggr == _lib.AGGR["mean"]:      # a node's 1 / deg once per node here, not once per edge in the launch (where it would hang
            dagg = rows_scale(dagg, ctx.handle.inv_in_degree(dev))   # on the edge's target index: one more dependent load per pass)
            aggr = _lib.AGGR["+"]
        dagg = dagg.contiguous()
        stream = _lib.current_stream()
        dT = torch.empty_like(T) if ctx.needs_input_grad[3] else None
        dBh = torch.empty((T.shape[0], cout), dtype=torch.float32, device=dev) if (has_bh and ctx.needs_input_grad[4]) else None
        want_p = pshape is not None and ctx.needs_input_grad[0]
        want_q = qshape is not None and ctx.needs_input_grad[1]
        want_e = has_e and ctx.needs_input_grad[2]
        dz = torch.empty_like(a) if (want_p or want_q or want_e) else None     # gradient of the pre-activation P[t] + Q[s] + E
        dQ = torch.empty(qshape, dtype=torch.float32, device=dev) if want_q else None
        _lib.check(lib.ngpde_gno_message_backward_from_nodes(ctx.handle.ptr, cout, kdim, aggr, act1, _lib.ptr(T), _lib.ptr(a),
                                                             _lib.ptr(dagg), _lib.ptr(dT), _lib.ptr(dBh), _lib.ptr(dz), _lib.ptr(dQ), stream))
        dP = None
        if want_p:                                                               # dP = sums of dz by target
            dP = torch.empty(pshape, dtype=torch.float32, device=dev)
            _lib.check(lib.ngpde_segment_reduce_forward(ctx.handle.ptr, kdim, _lib.AGGR["+"], _lib.ptr(dz), _lib.ptr(dP), stream))
        return (dP, dQ, (dz if want_e else None), dT, dBh) + (None,) * 7


def gno_message_aggregate(P, Q, Eterm, T, Bh, handle, act1, cout, kdim, n_edges, aggr, n_nodes):
    """Fused message + sum / mean aggregation (aggr: name or code); other aggregations: gno_message + segment_reduce."""
    code = _lib.AGGR[aggr] if isinstance(aggr, str) else int(aggr)
    if code not in (_lib.AGGR["+"], _lib.AGGR["mean"]):
        return segment_reduce(gno_message(P, Q, Eterm, T, Bh, handle, act1, cout, kdim, n_edges), handle, code, n_nodes)
    return _GnoMessageAggFn.apply(P, Q, Eterm, T, Bh, handle, int(act1), int(cout), int(kdim), int(n_edges), code, int(n_nodes))


def gno_message_supported(cout, kdim, act1):
    return act1 in (0, 1) and bool(_lib.load().ngpde_gno_message_supported(int(cout), int(kdim)))


def gno_message(P, Q, Eterm, T, Bh, handle, act1, cout, kdim, n_edges):
    return _GnoMessageFn.apply(P, Q, Eterm, T, Bh, handle, int(act1), int(cout), int(kdim), int(n_edges))


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


class _EdgeMlpFusedFn(torch.autograd.Function):
    """m_i = aggr_e phi(...) in one launch (ngpde_edge_mlp_forward): gather through LDS, MFMA layers, in-tile
    segmented reduction.  Training keeps the per-edge pre-activations and the pullback runs on the primitives."""

    @staticmethod
    def forward(ctx, P, Q, Eterm, handle, act1, aggr, n_nodes, n_edges, acts, *wb):
        lib = _lib.load()
        _need_cuda(P, Q, Eterm, *[t for t in wb if t is not None])
        ref = next(t for t in (P, Q, Eterm) if t is not None)
        dev, h1 = ref.device, ref.shape[1]
        P = None if P is None else P.contiguous()
        Q = None if Q is None else Q.contiguous()
        Eterm = None if Eterm is None else Eterm.contiguous()
        wts = [w.contiguous() for w in wb[0::2]]
        bs = list(wb[1::2])
        n_tail = len(wts)
        douts = [w.shape[1] for w in wts]
        need = any(ctx.needs_input_grad)
        widths = [h1] + douts
        # fused pullback available: it recomputes the per-edge activations, so the forward saves nothing per edge
        fused_bwd = need and os.environ.get("NGPDE_NO_FUSED_EDGE_BWD") != "1" and bool(lib.ngpde_edge_mlp_backward_supported(
            handle.ptr, h1, n_tail, _int_array(douts) if n_tail else None, aggr))
        if fused_bwd and n_tail >= 2:
            # message MLPs of three / four layers (edge_mlp_deep_bwd.hip): one 4-wave workgroup per CU walks a long dependent chain per
            # tile, which pays where a workgroup has many tiles to amortise it over -- 262 144 nodes: 3.4 against 4.7 ms forward +
            # backward; 3 000 nodes (the VMH tutorial): 105 us for the one launch against ~60 us of primitives' launches spread over
            # the whole chip (tools/bench_deep_mlp.py, tools/bench_vmh_node.py).  NGPDE_DEEP_EDGE_BWD=1 / 0 forces it on / off.
            force = os.environ.get("NGPDE_DEEP_EDGE_BWD")
            fused_bwd = force == "1" or (force != "0" and n_nodes >= 32768)
        saves = [torch.empty((n_edges, w), dtype=torch.float32, device=dev) if (need and not fused_bwd) else None for w in widths]
        out = torch.empty((n_nodes, widths[-1]), dtype=torch.float32, device=dev)
        _lib.check(lib.ngpde_edge_mlp_forward(handle.ptr, h1, act1, _lib.ptr(P), _lib.ptr(Q), _lib.ptr(Eterm), n_tail,
                                              _int_array(douts) if n_tail else None, _int_array(acts) if n_tail else None,
                                              _ptr_array(wts) if n_tail else None, _ptr_array(bs) if n_tail else None,
                                              aggr, _lib.ptr(out), _ptr_array(saves), _lib.current_stream()))
        ctx.handle, ctx.meta = handle, (act1, aggr, n_nodes, n_edges, tuple(acts), h1, tuple(douts))
        ctx.shapes = (None if P is None else P.shape, None if Q is None else Q.shape, Eterm is not None,
                      [b is not None for b in bs])
        ctx.fused_bwd = fused_bwd
        if fused_bwd:
            ctx.present = (P is not None, Q is not None, Eterm is not None)
            ctx.save_for_backward(*wts, *[b for b in bs if b is not None], *[t for t in (P, Q, Eterm) if t is not None])
        else:
            ctx.save_for_backward(*wts, *[s for s in saves if s is not None])
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        act1, aggr, n_nodes, n_edges, acts, h1, douts = ctx.meta
        n_tail = len(douts)
        if ctx.fused_bwd:
            return _EdgeMlpFusedFn._fused_backward(ctx, dout)
        saved = ctx.saved_tensors
        wts, zs = saved[:n_tail], saved[n_tail:]
        pshape, qshape, has_e, has_b = ctx.shapes
        dev = dout.device
        stream = _lib.current_stream()
        widths = [h1] + list(douts)
        dM = torch.empty((n_edges, widths[-1]), dtype=torch.float32, device=dev)
        _lib.check(lib.ngpde_segment_reduce_backward(ctx.handle.ptr, widths[-1], aggr, None, None, _lib.ptr(dout.contiguous()),
                                                     _lib.ptr(dM), stream))
        grads_wb = [None] * (2 * n_tail)
        for l in range(n_tail, 0, -1):
            z_prev, z_l, wt = zs[l - 1], zs[l], wts[l - 1]
            a_prev = torch.empty_like(z_prev)
            _lib.check(lib.ngpde_activation_forward(z_prev.numel(), act1 if l == 1 else acts[l - 2], _lib.ptr(z_prev),
                                                    _lib.ptr(a_prev), stream))
            dwt = torch.empty_like(wt)
            db = torch.empty((douts[l - 1],), dtype=torch.float32, device=dev) if has_b[l - 1] else None
            da = torch.empty_like(a_prev)
            ws = _ws(lib.ngpde_dense_workspace_bytes(n_edges, widths[l - 1], douts[l - 1]), dev)
            _lib.check(lib.ngpde_dense_backward(n_edges, 1, _ptr_array([a_prev]), _int_array([widths[l - 1]]), _int_array([1]),
                                                douts[l - 1], acts[l - 1], _lib.ptr(wt), _lib.ptr(z_l), _lib.ptr(dM),
                                                _ptr_array([da]), _lib.ptr(dwt), _lib.ptr(db), _lib.ptr(ws), ws.numel(), stream))
            grads_wb[2 * (l - 1)], grads_wb[2 * (l - 1) + 1] = dwt, db
            dM = da
        dz = torch.empty_like(dM)
        dP = torch.empty(pshape, dtype=torch.float32, device=dev) if pshape is not None else None
        dQ = torch.empty(qshape, dtype=torch.float32, device=dev) if qshape is not None else None
        _lib.check(lib.ngpde_edge_combine_backward(ctx.handle.ptr, h1, act1, _lib.ptr(dM), _lib.ptr(zs[0]), _lib.ptr(dz),
                                                   _lib.ptr(dP), _lib.ptr(dQ), stream))
        return (dP, dQ, dz if has_e else None, None, None, None, None, None, None, *grads_wb)


def _edge_mlp_fused_backward(ctx, dout):
    """ngpde_edge_mlp_backward: one fused launch (+ slab reduce + by-source sum)"""
    lib = _lib.load()
    act1, aggr, n_nodes, n_edges, acts, h1, douts = ctx.meta
    n_tail = len(douts)
    pshape, qshape, has_e, has_b = ctx.shapes
    saved = list(ctx.saved_tensors)
    wts = saved[:n_tail]
    nb = sum(has_b)
    bs_present = saved[n_tail:n_tail + nb]
    bs, it = [], iter(bs_present)
    for hb in has_b:
        bs.append(next(it) if hb else None)
    rest = iter(saved[n_tail + nb:])
    P = next(rest) if ctx.present[0] else None
    Q = next(rest) if ctx.present[1] else None
    Eterm = next(rest) if ctx.present[2] else None
    dev = dout.device
    dout = dout.contiguous()
    dP = torch.empty(pshape, dtype=torch.float32, device=dev) if pshape is not None else None
    dQ = torch.empty(qshape, dtype=torch.float32, device=dev) if qshape is not None else None
    # the [E][h1] array dz1: not needed where the 64-wide kernel sums it by source inside its launch (no per-edge term, mesh-like halos)
    need_de = has_e or dQ is None or bool(lib.ngpde_edge_mlp_backward_needs_edge_buffer(
        ctx.handle.ptr, h1, act1, 0, n_tail, _int_array(list(douts)) if n_tail else None, _int_array(list(acts)) if n_tail else None, aggr))
    dE = torch.empty((n_edges, h1), dtype=torch.float32, device=dev) if need_de else None
    dwts = [torch.empty_like(w) for w in wts]
    dbs = [torch.empty((douts[l],), dtype=torch.float32, device=dev) if has_b[l] else None for l in range(n_tail)]
    ws = _ws(lib.ngpde_edge_mlp_backward_workspace_bytes(ctx.handle.ptr, h1, n_tail, _int_array(list(douts)) if n_tail else None), dev)
    _lib.check(lib.ngpde_edge_mlp_backward(ctx.handle.ptr, h1, act1, _lib.ptr(P), _lib.ptr(Q), _lib.ptr(Eterm), n_tail,
                                           _int_array(list(douts)) if n_tail else None, _int_array(list(acts)) if n_tail else None,
                                           _ptr_array(wts) if n_tail else None, _ptr_array(bs) if n_tail else None, aggr,
                                           _lib.ptr(dout), _lib.ptr(dP), _lib.ptr(dQ), _lib.ptr(dE),
                                           _ptr_array(dwts) if n_tail else None, _ptr_array(dbs) if n_tail else None,
                                           _lib.ptr(ws), ws.numel(), _lib.current_stream()))
    grads_wb = []
    for l in range(n_tail):
        grads_wb += [dwts[l], dbs[l]]
    return (dP, dQ, dE if has_e else None, None, None, None, None, None, None, *grads_wb)


_EdgeMlpFusedFn._fused_backward = staticmethod(_edge_mlp_fused_backward)


def edge_mlp_supported(handle, h1, tail_douts):
    lib = _lib.load()
    return bool(lib.ngpde_edge_mlp_supported(handle.ptr, int(h1), len(tail_douts), _int_array(tail_douts) if tail_douts else None))


def edge_mlp_fused(P, Q, Eterm, handle, act1, aggr, n_nodes, n_edges, tail):
    """tail: list of (wt [in][out], bias or None, act code) for the layers after the first."""
    wb = []
    for wt, b, _ in tail:
        wb += [wt, b]
    return _EdgeMlpFusedFn.apply(P, Q, Eterm, handle, act1, _lib.AGGR[aggr] if isinstance(aggr, str) else aggr, n_nodes,
                                 n_edges, tuple(a for _, _, a in tail), *wb)


# ---- layer-level entries (api_layers.hip): one call per layer, one per pullback ----------------------------------------------------


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
        ctx.ws = None          # (the workspace holds every saved activation: let it go with the node)
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
        ctx.ws = None
        return (None, None, None, None, None, None, None, None, *grads)


def gno_layer(handle, cin, cout, aggr, act, h, lwt, lb, phi, node_feat=None, edge_feat=None):
    """phi: list of (weight [in][out], bias or None, activation code); lwt [in][out], lb [out] or None.  Returns [N][out]."""
    tensors = [h, lwt, lb]
    for wt, b, _ in phi:
        tensors += [wt, b]
    return _GnoLayerFn.apply(handle, int(cin), int(cout), _lib.AGGR[aggr] if isinstance(aggr, str) else int(aggr), int(act),
                             {"node_feat": node_feat, "edge_feat": edge_feat}, tuple(a for _, _, a in phi), torch.is_grad_enabled(), *tensors)


# ---- weight-sized rearrangements as library launches (row_blocks.hip) ---------------------------------------------------------


class _RowBlocksFn(torch.autograd.Function):
    """Recombined row blocks of a [rows][width] weight: every output is a vertical stack of blocks, a block the signed sum of
    equally long row ranges of the source (ExplicitEdgeConv [wa; -wc], VMHConv [wa - wb; -wc], MPPDEConv [wa; wc; we] ...).
    ONE launch builds all outputs, ONE launch their pullback -- no slices, cat, neg, zero-fills or adds of slice gradients."""

    @staticmethod
    def forward(ctx, wt, spec):
        import ctypes as C
        lib = _lib.load()
        wt = wt.contiguous()
        rows, width = wt.shape
        out_index, dst0, src0, nrows, sign, out_rows = [], [], [], [], [], []
        for o, blocks in enumerate(spec):
            r = 0
            for n, terms in blocks:
                for s0, sg in terms:
                    out_index.append(o); dst0.append(r); src0.append(s0); nrows.append(n); sign.append(float(sg))
                r += n
            out_rows.append(r)
        outs = [torch.empty((r, width), dtype=torch.float32, device=wt.device) for r in out_rows]
        meta = (rows, width, _int_array(out_index), _int_array(dst0), _int_array(src0), _int_array(nrows),
                (C.c_float * max(len(sign), 1))(*sign), len(out_index), _int_array(out_rows), len(outs))
        _lib.check(lib.ngpde_row_blocks_gather(width, rows, _lib.ptr(wt), meta[7], meta[2], meta[3], meta[4], meta[5], meta[6], len(outs),
                                               _ptr_array(outs), meta[8], _lib.current_stream()))
        ctx.meta, ctx.dev = meta, wt.device
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        lib = _lib.load()
        rows, width, out_index, dst0, src0, nrows, sign, n_seg, out_rows, n_out = ctx.meta
        douts = [None if d is None else d.contiguous() for d in douts]
        dwt = torch.empty((rows, width), dtype=torch.float32, device=ctx.dev)
        _lib.check(lib.ngpde_row_blocks_scatter(width, rows, _lib.ptr(dwt), n_seg, out_index, dst0, src0, nrows, sign, n_out,
                                                _ptr_array(douts), out_rows, _lib.current_stream()))
        return dwt, None


def row_blocks(wt, spec):
    """spec: per output a list of blocks (n_rows, [(src_row0, sign), ...]); blocks with n_rows == 0 are dropped.  Returns the
    list of output matrices ([sum n_rows][width] each)."""
    spec = [[(int(n), list(terms)) for n, terms in blocks if n > 0] for blocks in spec]
    return list(_RowBlocksFn.apply(wt, spec))


class _TransposeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        a = a.contiguous()
        out = torch.empty((a.shape[1], a.shape[0]), dtype=torch.float32, device=a.device)
        _lib.check(_lib.load().ngpde_transpose(a.shape[0], a.shape[1], _lib.ptr(a), _lib.ptr(out), _lib.current_stream()))
        return out

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        out = torch.empty((d.shape[1], d.shape[0]), dtype=torch.float32, device=d.device)
        _lib.check(_lib.load().ngpde_transpose(d.shape[0], d.shape[1], _lib.ptr(d), _lib.ptr(out), _lib.current_stream()))
        return out


def transpose(a):
    """contiguous transpose of a 2-D float32 tensor as a library launch (its pullback: the transpose of the cotangent)"""
    return _TransposeFn.apply(a)


def rows_scale(x, scale):
    """out[i][:] = x[i][:] * scale[i] (no autograd: used inside pullbacks)"""
    x = x.contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.load().ngpde_rows_scale(x.shape[0], x.shape[1], _lib.ptr(x), _lib.ptr(scale), _lib.ptr(out), _lib.current_stream()))
    return out


# ---- several independent small Dense layers in one launch; fan-out of a tensor to several consumers ----------------------------


def _sum_list(tensors):
    """sum of equally shaped float32 tensors as ONE library launch (the Runge-Kutta combination kernel), no torch adds"""
    import ctypes as C
    tensors = [t.contiguous() for t in tensors]
    if len(tensors) == 1:
        return tensors[0]
    out = torch.empty_like(tensors[0])
    terms = tensors[1:]
    _lib.check(_lib.load().ngpde_rk_stage_combine(out.numel(), 1.0, _lib.ptr(tensors[0]), len(terms),
                                                  (C.c_void_p * len(terms))(*[t.data_ptr() for t in terms]),
                                                  (C.c_float * len(terms))(*([1.0] * len(terms))), _lib.ptr(out), _lib.current_stream()))
    return out


class _FanoutFn(torch.autograd.Function):
    """k aliases of one tensor for k consumers; the pullback sums their cotangents in ONE launch (autograd would add them pairwise
    with torch kernels)"""

    @staticmethod
    def forward(ctx, x, k):
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *ds):
        ds = [d for d in ds if d is not None]
        return (_sum_list(ds) if ds else None), None


def fanout(x, k):
    return list(_FanoutFn.apply(x, k)) if (k > 1 and x.requires_grad and torch.is_grad_enabled()) else [x] * k


class _DenseMultiFn(torch.autograd.Function):
    """Up to four independent Dense layers y_q = act_q(x_q W_q + b_q) (ONE input block each) in one launch
    (ngpde_dense_multi_forward); the pullbacks are the single-problem launches; inputs that are the same tensor get the sum."""

    @staticmethod
    def forward(ctx, acts, *args):          # args: x_0, wt_0, b_0, x_1, wt_1, b_1, ...
        import ctypes as C
        lib = _lib.load()
        q = len(acts)
        xs = [args[3 * i].contiguous() for i in range(q)]
        wts = [args[3 * i + 1].contiguous() for i in range(q)]
        bs = [args[3 * i + 2] for i in range(q)]
        _need_cuda(*xs, *wts, *bs)
        for x, wt in zip(xs, wts):
            if wt.shape[0] != x.shape[1]:
                raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                             f"DimensionMismatch: Dense expects {wt.shape[0]} input features, got {x.shape[1]}")
        dev = wts[0].device
        need = any(ctx.needs_input_grad)
        ys = [torch.empty((x.shape[0], wt.shape[1]), dtype=torch.float32, device=dev) for x, wt in zip(xs, wts)]
        zs = [torch.empty_like(y) if (need and a != 0) else None for y, a in zip(ys, acts)]
        n = (C.c_int64 * q)(*[x.shape[0] for x in xs])
        _lib.check(lib.ngpde_dense_multi_forward(q, n, _int_array([1] * q), _ptr_array(xs), _int_array([x.shape[1] for x in xs]),
                                                 _int_array([1] * q), _int_array([wt.shape[1] for wt in wts]), _int_array(acts),
                                                 _ptr_array(wts), _ptr_array(bs), _ptr_array(ys), _ptr_array(zs), _lib.current_stream()))
        ctx.acts = tuple(acts)
        ctx.has_b = tuple(b is not None for b in bs)
        ctx.same = [next(j for j in range(q) if args[3 * j] is args[3 * i]) for i in range(q)]   # first problem with the same input
        ctx.save_for_backward(*xs, *wts, *[z if z is not None else torch.empty(0, device=dev) for z in zs])
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        lib = _lib.load()
        q = len(ctx.acts)
        saved = ctx.saved_tensors
        xs, wts, zs = saved[:q], saved[q:2 * q], saved[2 * q:]
        grads = [None] * (3 * q)
        dxs = [None] * q
        for i in range(q):
            if dys[i] is None:
                continue
            want_x = ctx.needs_input_grad[1 + 3 * i]
            z = zs[i] if zs[i].numel() else None
            dwt, db, dblocks = _dense_backward_call(lib, xs[i].shape[0], [xs[i]], [xs[i].shape[1]], [1], wts[i].shape[1], ctx.acts[i], wts[i],
                                                    z, dys[i].contiguous(), [want_x], ctx.has_b[i])
            grads[3 * i + 1], grads[3 * i + 2], dxs[i] = dwt, db, dblocks[0]
        for i in range(q):                                   # one gradient per distinct input tensor
            if ctx.same[i] != i or not ctx.needs_input_grad[1 + 3 * i]:
                continue
            parts = [dxs[j] for j in range(q) if ctx.same[j] == i and dxs[j] is not None]
            grads[3 * i] = _sum_list(parts) if parts else None
        return (None, *grads)


def dense_multi(problems):
    """problems: list of (x [n][k], wt [k][m], bias or None, act code); returns the list of outputs.  Empty inputs are allowed
    (zero rows)."""
    args = []
    for x, wt, b, _ in problems:
        args += [x, wt, b]
    return list(_DenseMultiFn.apply([int(a) for *_, a in problems], *args))

"""Differentiable wrappers over the C ABI (the Python stand-in for the `ChainRulesCore.rrule`s the
Julia shim of INTEGRATION.md defines).  All tensors here are in KERNEL layout: features [N][D]
row-major float32 on the GPU (= a Julia (D x N) matrix), weights [in][out] (= Julia (out x in)).
"""
from __future__ import annotations

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
    """y = act(W (x C (A+I) C) + b)  -- /root/reference/src/layers.jl:200-239."""

    @staticmethod
    def forward(ctx, x, wt, bias, handle, act):
        lib = _lib.load()
        _need_cuda(x, wt, bias)
        n, din = x.shape
        dout = wt.shape[1]
        if wt.shape[0] != din:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                         f"DimensionMismatch: weight is ({dout} x {wt.shape[0]}), x has {din} features")
        x = x.contiguous()
        wt = wt.contiguous()
        need_grad = any(ctx.needs_input_grad[:3])
        y = torch.empty((n, dout), dtype=torch.float32, device=x.device)
        agg = torch.empty((n, din), dtype=torch.float32, device=x.device) if (need_grad and dout >= din) else None
        z = torch.empty((n, dout), dtype=torch.float32, device=x.device) if need_grad else None
        ws = _ws(lib.ngpde_gcn_workspace_bytes(handle.ptr, din, dout, 0), x.device)
        _lib.check(lib.ngpde_gcn_forward(handle.ptr, din, dout, act, _lib.ptr(x), _lib.ptr(wt), _lib.ptr(bias),
                                         _lib.ptr(y), _lib.ptr(agg), _lib.ptr(z), _lib.ptr(ws), ws.numel(),
                                         _lib.current_stream()))
        ctx.handle, ctx.act, ctx.dims = handle, act, (n, din, dout)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, wt, z, agg)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, wt, z, agg = ctx.saved_tensors
        n, din, dout = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dwt = torch.empty_like(wt)
        db = torch.empty((dout,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
        ws = _ws(lib.ngpde_gcn_workspace_bytes(ctx.handle.ptr, din, dout, 1), x.device)
        _lib.check(lib.ngpde_gcn_backward(ctx.handle.ptr, din, dout, ctx.act, _lib.ptr(x), _lib.ptr(wt), _lib.ptr(z),
                                          _lib.ptr(agg), _lib.ptr(dy), _lib.ptr(dx), _lib.ptr(dwt), _lib.ptr(db),
                                          _lib.ptr(ws), ws.numel(), _lib.current_stream()))
        return dx, dwt, db, None, None


def gcn_conv(x, wt, bias, handle, act):
    return _GCNConvFn.apply(x, wt, bias, handle, act)


def propagate_copy_xj(x, handle, aggr="+", edge_weight=None, by_source=False):
    """out[i] = aggr_{e: t_e = i} w_e x[s_e]   (no autograd; forward primitive)."""
    lib = _lib.load()
    _need_cuda(x, edge_weight)
    x = x.contiguous()
    out = torch.empty_like(x)
    _lib.check(lib.ngpde_propagate_copy_xj(handle.ptr, x.shape[1], _lib.AGGR[aggr], int(by_source), _lib.ptr(x),
                                           _lib.ptr(edge_weight), _lib.ptr(out), _lib.current_stream()))
    return out

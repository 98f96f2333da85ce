"""Batches of point clouds as one solver state (docs/src/tutorials/VMH.md:120-134 of the reference: a DataLoader hands the training loop
block-diagonal batches of single graphs): the padded numbering a device-resident plan needs when the members are not whole 32-row tiles
(the arithmetic is the library's, ngpde_batch_pad_host), and the re-use of an earlier batch's graph, plan and tapes when an epoch brings
the same clouds in another order."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

import torch

from . import _lib


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
    poff, index = np.zeros(len(members) + 1, dtype=np.int64), np.zeros(int(sizes.sum()), dtype=np.int64)
    order = g._shared.get("order")
    order = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
    order_p = None if order is None else np.zeros(int(((sizes + 31) // 32 * 32).sum()), dtype=np.int32)
    np_ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)   # noqa: E731
    _lib.check(_lib.load().ngpde_batch_pad_host(len(members), np_ptr(sizes), np_ptr(poff), np_ptr(index), np_ptr(order), np_ptr(order_p)))
    s0, t0 = g.edge_index(index_base=0)
    s0, t0 = np.asarray(s0.cpu() if isinstance(s0, torch.Tensor) else s0), np.asarray(t0.cpu() if isinstance(t0, torch.Tensor) else t0)
    gp = GNNGraph(index[s0], index[t0], num_nodes=int(poff[-1]), index_base=0, num_graphs=len(members))
    x = _as_matrix_t(g.ndata["x"], g.num_nodes).to(device)                      # [N][pd]
    idx_t = torch.as_tensor(index, device=device)
    xp = torch.zeros((int(poff[-1]), x.shape[1]), dtype=torch.float32, device=device).index_copy(0, idx_t, x.to(torch.float32))
    gp.ndata = {"x": xp.T}
    if order_p is not None:      # the members' locality orders, each followed by its padding nodes
        gp._shared["order"] = order_p
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

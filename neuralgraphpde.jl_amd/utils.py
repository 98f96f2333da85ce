"""State utilities -- mirror of /root/reference/src/utils.jl."""
from __future__ import annotations

from .graphs import GNNGraph


def drop(nt: dict, key):
    """src/utils.jl:1  drop(nt, key) = structdiff(nt, (key,))"""
    return {k: v for k, v in nt.items() if k != key}


def wrapgraph(g):
    """src/utils.jl:16-17: a graph becomes the thunk () -> copy(g); functions pass through."""
    if isinstance(g, GNNGraph):
        return lambda: g.copy()
    if callable(g):
        return g
    raise TypeError(f"initialgraph must be a GNNGraph or a function returning one, got {type(g)}")


def updategraph(st: dict, g=None, **kwargs):
    """src/utils.jl:24-31: recursively replace every GNNGraph leaf of `st` by `g` (the same object,
    as the reference's test asserts `new_st.graph === new_g`, test/runtests.jl:179), or -- when g is
    None -- by a copy of the old graph with the data given in kwargs (ndata / edata / gdata)."""
    if not st:
        return st
    out = {}
    for k, v in st.items():
        if isinstance(v, GNNGraph):
            out[k] = g if g is not None else v.copy(**kwargs)
        elif isinstance(v, dict):
            out[k] = updategraph(v, g, **kwargs)
        else:
            out[k] = v
    return out

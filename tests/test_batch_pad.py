"""`ngpde_batch_pad_host` on the CPU (no GPU call): the numbering of a block-diagonal batch (test/runtests.jl:89-102 of the reference,
VMH.md:120-134) whose members are padded to whole 32-row tiles -- offsets, the gather / scatter index of the real nodes and the padded
locality order -- against the same thing written with numpy; refusals of malformed input."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng  # noqa: E402,F401
from ngpde_amd import _lib  # noqa: E402

ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731


def pad(sizes, order=None):
    sizes = np.asarray(sizes, dtype=np.int64)
    poff = np.full(len(sizes) + 1, -1, dtype=np.int64)
    index = np.full(int(sizes.sum()), -1, dtype=np.int64)
    n_pad = int(((sizes + 31) // 32 * 32).sum())
    order_p = None if order is None else np.full(n_pad, -1, dtype=np.int32)
    order = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
    rc = _lib.load().ngpde_batch_pad_host(len(sizes), ptr(sizes) if len(sizes) else None, ptr(poff), ptr(index) if index.size else None,
                                          ptr(order), ptr(order_p))
    return rc, poff, index, order_p


@pytest.mark.parametrize("sizes", [[3000, 2999, 31, 32, 33, 1], [64, 64], [77], [0, 5, 0]])
def test_offsets_index_and_order_against_numpy(sizes):
    sizes = np.asarray(sizes, dtype=np.int64)
    rng = np.random.default_rng(int(sizes.sum()))
    off = np.concatenate([[0], np.cumsum(sizes)])
    order = np.concatenate([off[k] + rng.permutation(int(n)) for k, n in enumerate(sizes)]).astype(np.int32)
    rc, poff, index, order_p = pad(sizes, order)
    assert rc == 0, _lib.load().ngpde_last_error().decode()
    padded = (sizes + 31) // 32 * 32
    want_poff = np.concatenate([[0], np.cumsum(padded)])
    assert np.array_equal(poff, want_poff)
    want_index = np.concatenate([np.arange(n, dtype=np.int64) + po for n, po in zip(sizes, want_poff[:-1])]) if sizes.sum() else np.zeros(0, np.int64)
    assert np.array_equal(index, want_index)
    parts = []
    for k in range(len(sizes)):
        parts.append(want_index[order[off[k]:off[k + 1]].astype(np.int64)])
        parts.append(np.arange(want_poff[k] + sizes[k], want_poff[k + 1], dtype=np.int64))
    assert np.array_equal(order_p, np.concatenate(parts).astype(np.int32))
    assert np.array_equal(np.sort(order_p), np.arange(want_poff[-1]))          # a permutation of the padded nodes
    # every member starts on a tile boundary and its padding nodes are exactly the ones no real node maps to
    assert np.all(want_poff % 32 == 0)
    assert np.setdiff1d(np.arange(want_poff[-1]), index).size == int((padded - sizes).sum())


def test_without_an_order_and_with_no_members():
    rc, poff, index, order_p = pad([40, 8])
    assert rc == 0 and order_p is None and list(poff) == [0, 64, 96] and list(index[38:42]) == [38, 39, 64, 65]
    rc, poff, _, _ = pad([])
    assert rc == 0 and list(poff) == [0]


def test_refusals():
    lib = _lib.load()
    rc, *_ = pad([5, -1])
    assert rc == _lib.ERR["INVALID_ARGUMENT"] if hasattr(_lib, "ERR") else rc != 0
    assert "member 1" in lib.ngpde_last_error().decode()
    rc, *_ = pad([4, 4], order=[0, 1, 2, 4, 3, 5, 6, 7])                       # node 4 listed inside member 0
    assert rc != 0 and "not a permutation inside member 0" in lib.ngpde_last_error().decode()
    rc, *_ = pad([4], order=[0, 1, 1, 3])                                      # a node twice
    assert rc != 0
    sizes = np.asarray([4], dtype=np.int64)
    assert lib.ngpde_batch_pad_host(1, ptr(sizes), None, None, None, None) != 0   # padded_offsets is required

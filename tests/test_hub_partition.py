"""Host logic of the persistent solver's hub geometry, on the CPU: `ngpde_hub_partition_host` returns the tile partition
`node_persistent_setup` would use for a graph with hubs -- a Cora-shaped graph (docs/src/tutorials/graph_node.md:14-23 of the reference:
2 708 nodes, 5 278 pairs), stars around the cap of ~ 224 distinct neighbours, directed hubs, random graphs -- and its invariants are
recomputed here with numpy: every node in exactly one row, every tile within 256 referenced rows in both directions, the hubs dealt one
per tile.  No GPU call."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng  # noqa: E402,F401
from ngpde_amd import _lib, synth as S  # noqa: E402

TILE, HALO = 32, 256


def csr(n, rows, cols):
    order = np.argsort(rows, kind="stable")
    rp = np.zeros(n + 1, dtype=np.int32)
    np.add.at(rp, rows + 1, 1)
    return np.cumsum(rp, dtype=np.int32), cols[order].astype(np.int32)


def partition(n, s, t):
    lib = _lib.load()
    f = lib.ngpde_hub_partition_host          # (signature: ngpde_amd/_lib.py)
    s, t = np.asarray(s, dtype=np.int64), np.asarray(t, dtype=np.int64)
    rpt, clt = csr(n, t, s)          # by target: in-neighbours
    rps, cls = csr(n, s, t)          # by source: out-neighbours
    nt = (n + TILE - 1) // TILE
    order = np.full(n, -1, dtype=np.int32)
    rows = np.zeros((nt, 2), dtype=np.int32)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None  # noqa: E731
    rc = f(n, ptr(rpt), ptr(clt), ptr(rps), ptr(cls), ptr(order), ptr(rows))
    return rc, order, rows, (rpt, clt, rps, cls), lib.ngpde_last_error().decode()


def check(n, order, rows, lists):
    rpt, clt, rps, cls = lists
    assert sorted(order.tolist()) == list(range(n))                       # every node in exactly one row
    nt = (n + TILE - 1) // TILE
    for tl in range(nt):
        members = order[TILE * tl: TILE * (tl + 1)]
        for d, (rp, cl) in enumerate(((rpt, clt), (rps, cls))):
            ref = set(members.tolist())
            for v in members:
                ref.update(cl[rp[v]:rp[v + 1]].tolist())
            assert len(ref) <= HALO
            # the library counts the slots it hands out: the tile's 32 row slots (padding included) + the foreign rows
            assert rows[tl, d] == len(ref) + (TILE - members.size)


def symmetric(s, t):
    return np.concatenate([s, t]), np.concatenate([t, s])


def test_cora_shaped_graph_partitions_with_hubs_dealt_apart():
    n = 2708
    s, t = S.preferential_pairs_graph(n, 5278, seed=1)
    rc, order, rows, lists, msg = partition(n, s, t)
    assert rc == 0, msg
    check(n, order, rows, lists)
    deg = np.bincount(t, minlength=n)
    nt = (n + TILE - 1) // TILE
    assert deg.max() > 100
    hubs = np.argsort(-deg, kind="stable")[:nt]                          # the n_tiles highest-degree nodes: one per tile
    tile_of = np.empty(n, dtype=np.int64)
    tile_of[order] = np.arange(n) // TILE
    assert len(set(tile_of[hubs].tolist())) == nt
    assert rows.max() <= HALO and rows[:, 0].max() < 200                 # balanced: no tile near the cap on this graph


@pytest.mark.parametrize("hub_degree,fits", [(40, True), (200, True), (224, True), (225, False), (300, False)])
def test_star_around_the_neighbour_cap(hub_degree, fits):
    # a hub of `hub_degree` distinct neighbours (its two ring neighbours + leaves) in a ring of 1 024 nodes: the hub's tile holds the hub,
    # its neighbours' rows and one row reserved for each of its 31 other members -- 1 + 224 + 31 = 256 is the last that fits
    n = 1024
    ring = np.arange(n)
    s, t = symmetric(ring, (ring + 1) % n)
    leaves = np.arange(2, hub_degree)                                     # + nodes 1 and n - 1 of the ring
    hs, ht = symmetric(np.zeros(leaves.size, dtype=np.int64), leaves)
    s, t = np.concatenate([s, hs]), np.concatenate([t, ht])
    rc, order, rows, lists, msg = partition(n, s, t)
    if fits:
        assert rc == 0, msg
        check(n, order, rows, lists)
    else:
        assert rc == _lib.ERR_UNSUPPORTED and "224" in msg, (rc, msg)


def test_directed_hub_counts_both_directions():
    # 150 edges INTO node 0 and 150 different edges OUT of it: 300 distinct neighbours in the union -> no tile can hold the hub
    n = 2048
    ring = np.arange(n)
    s, t = symmetric(ring, (ring + 1) % n)
    into, out = np.arange(10, 160), np.arange(400, 550)
    s2 = np.concatenate([s, into, np.zeros(150, dtype=np.int64)])
    t2 = np.concatenate([t, np.zeros(150, dtype=np.int64), out])
    rc, _, _, _, msg = partition(n, s2, t2)
    assert rc == _lib.ERR_UNSUPPORTED, msg
    # 100 + 100: fits, and the two directions' row counts differ in the hub's tile
    s3 = np.concatenate([s, into[:100], np.zeros(100, dtype=np.int64)])
    t3 = np.concatenate([t, np.zeros(100, dtype=np.int64), out[:100]])
    rc, order, rows, lists, msg = partition(n, s3, t3)
    assert rc == 0, msg
    check(n, order, rows, lists)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_random_graphs_with_hubs(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([200, 700, 1500, 4000, 8000]))
    s, t = S.preferential_pairs_graph(n, int(n * rng.uniform(1.2, 2.5)), seed=seed)
    rc, order, rows, lists, msg = partition(n, s, t)
    union = [len(set(lists[1][lists[0][v]:lists[0][v + 1]].tolist()) | set(lists[3][lists[2][v]:lists[2][v + 1]].tolist()) - {v}) for v in range(n)]
    if rc == 0:
        check(n, order, rows, lists)
    else:                                                                 # refused: only for the documented reason
        assert rc == _lib.ERR_UNSUPPORTED and max(union) > 224 - TILE, (rc, msg, max(union))


def test_edgeless_and_ragged_graphs_and_bad_arguments():
    rc, order, rows, lists, msg = partition(70, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
    assert rc == 0, msg
    check(70, order, rows, lists)
    assert rows[:2].tolist() == [[TILE, TILE]] * 2 and rows[2].tolist() == [TILE, TILE]      # own row slots only
    lib = _lib.load()
    f = lib.ngpde_hub_partition_host          # (signature: ngpde_amd/_lib.py)
    assert f(0, None, None, None, None, None, None) == _lib.ERR_INVALID_ARGUMENT
    rp = np.array([0, 1, 1], dtype=np.int32)
    bad = np.array([7], dtype=np.int32)                                    # names node 7 of 2
    order = np.zeros(2, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert f(2, p(rp), p(bad), p(rp), p(bad), p(order), None) == _lib.ERR_INVALID_ARGUMENT
    # more than 256 tiles: not the hub geometry's case
    n = 32 * 257
    rp0 = np.zeros(n + 1, dtype=np.int32)
    order = np.zeros(n, dtype=np.int32)
    assert f(n, p(rp0), None, p(rp0), None, p(order), None) == _lib.ERR_UNSUPPORTED

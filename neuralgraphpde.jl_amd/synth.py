"""synth.py -- portable synthetic inputs for tests and bench (numpy only, no GPU needed).

Counter-based splitmix64 streams, so this container and the GPU box regenerate identical large
inputs from a seed instead of shipping them: the BASELINE config-2 graph
(16384 uniform points, the 65536 closest pairs as 131072 symmetric directed edges), N(0,1)
features and glorot-uniform weights.
"""
from __future__ import annotations

import numpy as np


_M64 = (1 << 64) - 1


def splitmix64(seed, n):
    """n uint64 values of the splitmix64 stream started at `seed` (vectorised)."""
    idx = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed & _M64)
    z = idx
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(seed, n):
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal(seed, n):
    u1 = uniform01(seed, n)
    u2 = uniform01(seed ^ 0x5DEECE66D, n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def closest_pairs_graph(n_nodes, n_pairs, seed):
    """C2 graph: n_nodes uniform points in [0,1]^2, the n_pairs closest pairs as symmetric
    directed edges (E = 2 n_pairs).  Deterministic given seed (cell-grid candidate search +
    stable sort on (distance, i, j))."""
    pts = np.stack([uniform01(seed, n_nodes), uniform01(seed + 1, n_nodes)], axis=1)
    # candidate radius: expected pairs within r is n^2/2 * pi r^2 -> take 1.6x the target
    r = np.sqrt(1.6 * n_pairs * 2.0 / (np.pi * n_nodes * n_nodes))
    while True:
        ncell = max(1, int(1.0 / r))
        cx = np.minimum((pts[:, 0] * ncell).astype(np.int64), ncell - 1)
        cy = np.minimum((pts[:, 1] * ncell).astype(np.int64), ncell - 1)
        cell = cx * ncell + cy
        order = np.argsort(cell, kind="stable")
        start = np.searchsorted(cell[order], np.arange(ncell * ncell + 1))
        I, J = [], []
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                nx, ny = cx + dx, cy + dy
                ok = (nx >= 0) & (nx < ncell) & (ny >= 0) & (ny < ncell)
                nc = np.where(ok, nx * ncell + ny, 0)
                lo, hi = start[nc], np.where(ok, start[nc + 1], start[nc])
                cnt = hi - lo
                src = np.repeat(np.arange(n_nodes), cnt)
                offs = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
                dst = order[np.repeat(lo, cnt) + offs]
                keep = src < dst
                I.append(src[keep])
                J.append(dst[keep])
        I, J = np.concatenate(I), np.concatenate(J)
        d2 = ((pts[I] - pts[J]) ** 2).sum(axis=1)
        within = d2 <= r * r
        I, J, d2 = I[within], J[within], d2[within]
        if I.size >= n_pairs:
            break
        r *= 1.3
    o = np.lexsort((J, I, d2))[:n_pairs]
    I, J = I[o], J[o]
    s = np.concatenate([I, J])
    t = np.concatenate([J, I])
    return pts, s.astype(np.int64), t.astype(np.int64)


def glorot_uniform(seed, out_dims, in_dims):
    lim = np.sqrt(6.0 / (in_dims + out_dims))
    return ((uniform01(seed, out_dims * in_dims) * 2.0 - 1.0) * lim).reshape(out_dims, in_dims)


def preferential_pairs_graph(n_nodes, n_pairs, seed, exponent=0.5):
    """C1 graph: n_pairs distinct undirected pairs (no self loops) as symmetric directed edges, drawn with a
    preferential-attachment bias -- endpoint i with probability ~ (i + 1)^-exponent -- so that degrees are skewed the way
    Cora's are (Cora itself is not on disk): at 2 708 nodes / 5 278 pairs the largest degree is ~100, the median 3.
    Deterministic given seed (splitmix64 streams; duplicates and self pairs are dropped in draw order)."""
    w = (np.arange(n_nodes, dtype=np.float64) + 1.0) ** (-exponent)
    cdf = np.cumsum(w) / w.sum()
    chosen, seen, rnd = [], set(), 0
    while len(chosen) < n_pairs:
        m = 2 * n_pairs
        a = np.minimum(np.searchsorted(cdf, uniform01(seed + 7919 * rnd, m)), n_nodes - 1)
        b = np.minimum(np.searchsorted(cdf, uniform01(seed + 7919 * rnd + 1, m)), n_nodes - 1)
        for i, j in zip(a.tolist(), b.tolist()):
            if i == j:
                continue
            key = (min(i, j), max(i, j))
            if key in seen:
                continue
            seen.add(key)
            chosen.append(key)
            if len(chosen) == n_pairs:
                break
        rnd += 2
    pa = np.array(chosen, dtype=np.int64)
    return np.concatenate([pa[:, 0], pa[:, 1]]), np.concatenate([pa[:, 1], pa[:, 0]])


def periodic_mesh_batch(n, traj, reach=3):
    """C4 graph: `traj` copies of an n-node periodic 1-D mesh, every node linked to its `reach` neighbours on each side
    (2 * reach * n directed edges per trajectory), as one block-diagonal COO list (graph by graph)."""
    idx = np.arange(n)
    offs = [k for k in range(-reach, reach + 1) if k != 0]
    s = np.concatenate([idx for _ in offs])
    t = np.concatenate([(idx + k) % n for k in offs])
    return (np.concatenate([s + i * n for i in range(traj)]).astype(np.int64),
            np.concatenate([t + i * n for i in range(traj)]).astype(np.int64))


def grid_radius_graph(k, radius):
    """C5 graph: the k x k cell-centred grid on [0, 1]^2, every node linked to the nodes strictly within `radius` (no self
    loops).  Returns (points (2 x k^2), s, t), edges grouped by offset."""
    gx, gy = np.meshgrid((np.arange(k) + 0.5) / k, (np.arange(k) + 0.5) / k, indexing="ij")
    pts = np.stack([gx.ravel(), gy.ravel()])
    cell = int(np.ceil(radius * k)) + 1
    ii, jj = np.divmod(np.arange(k * k), k)
    ss, tt = [], []
    for di in range(-cell, cell + 1):
        for dj in range(-cell, cell + 1):
            if di == 0 and dj == 0:
                continue
            ni, nj = ii + di, jj + dj
            ok = (ni >= 0) & (ni < k) & (nj >= 0) & (nj < k) & ((di / k) ** 2 + (dj / k) ** 2 < radius ** 2)
            ss.append((ni * k + nj)[ok])
            tt.append(np.arange(k * k)[ok])
    return pts, np.concatenate(ss).astype(np.int64), np.concatenate(tt).astype(np.int64)

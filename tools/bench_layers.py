"""Secondary measurements (not the bench.py metric): one layer forward+backward of BASELINE configs
C3 (GATConv 4x16 on the C2 graph), C4 (MPPDEConv, per-GPU shard of 64 trajectories x 8192-node periodic
mesh, h = 64; optionally fewer trajectories) and C5 (GNOConv on a 64x64 grid radius graph, width W).
Prints one JSON line per config with ms fwd / ms fwd+bwd."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

DEV = "cuda"


def timeit(fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def grad_leaves(ps):
    out = []
    for v in ps.values():
        out += grad_leaves(v) if isinstance(v, dict) else [v]
    return out


def colmajor(d, n):
    """(d x n) features in the reference's memory layout (Julia column-major == row-major [n][d]): no transposition copy"""
    return torch.randn(n, d, device=DEV).T


def run(name, layer, x, ps, st, reps, extra):
    ps = ng.to_device(ps, DEV)
    for v in grad_leaves(ps): v.requires_grad_(True)
    x = x.detach().requires_grad_(True)
    fwd = lambda: layer(x, ps, st)[0]
    with torch.no_grad():
        y0 = fwd()
    R = colmajor(y0.shape[0], y0.shape[1])          # cotangent in the same layout as the output
    def fb():
        y = layer(x, ps, st)[0]
        y.backward(R)
    with torch.no_grad():
        ms_f = timeit(fwd, reps)
    ms_fb = timeit(fb, reps)
    print(json.dumps(dict(config=name, ms_forward=round(ms_f, 3), ms_forward_backward=round(ms_fb, 3), **extra)), flush=True)


def c3(reps):
    _, s, t = S.closest_pairs_graph(16384, 65536, seed=2)
    g = ng.GNNGraph(s, t, num_nodes=16384, index_base=0)
    l = ng.GATConv((64, 16), "relu", heads=4, initialgraph=g)
    ps, st = ng.setup(3, l)
    run("C3 GATConv 64=>4x16, 16384 nodes / 131072 edges + self loops", l, colmajor(64, 16384), ps, st, reps,
        dict(nodes=16384, edges=131072))


def c4(reps, traj, act="swish"):
    n, h = 8192, 64
    idx = np.arange(n)
    s = np.concatenate([idx for k in (-3, -2, -1, 1, 2, 3)])
    t = np.concatenate([(idx + k) % n for k in (-3, -2, -1, 1, 2, 3)])
    S_, T_ = np.concatenate([s + i * n for i in range(traj)]), np.concatenate([t + i * n for i in range(traj)])
    N = n * traj
    rng = np.random.default_rng(4)
    g = ng.GNNGraph(S_, T_, num_nodes=N, index_base=0, num_graphs=traj,
                    ndata={"u": torch.rand(1, N), "x": torch.as_tensor(np.tile(idx / n, traj)[None, :].astype(np.float32))},
                    gdata={"θ": torch.rand(2, traj)})
    phi = ng.Chain(ng.Dense(132, 64, act), ng.Dense(64, 64, act))
    psi = ng.Chain(ng.Dense(130, 64, act), ng.Dense(64, 64))
    l = ng.MPPDEConv(phi, psi, initialgraph=g)
    ps, st = ng.setup(4, l)
    run(f"C4 MPPDEConv h=64, {traj} trajectories x 8192-node periodic mesh (6 neighbours)", l,
        colmajor(h, N), ps, st, reps, dict(nodes=N, edges=int(S_.size), trajectories=traj))


def c5(reps, width, radius):
    gx, gy = np.meshgrid(np.linspace(0, 1, 64), np.linspace(0, 1, 64), indexing="ij")
    pts = np.stack([gx.ravel(), gy.ravel()], 1)
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    s, t = np.nonzero((d2 <= radius * radius) & ~np.eye(4096, dtype=bool))
    g = ng.GNNGraph(s, t, num_nodes=4096, index_base=0,
                    ndata={"a": torch.rand(1, 4096), "x": torch.as_tensor(pts.T.astype(np.float32))})
    phi = ng.Chain(ng.Dense(6, 64, "relu"), ng.Dense(64, width * width))
    l = ng.GNOConv((width, width), phi, "relu", initialgraph=g)
    ps, st = ng.setup(5, l)
    run(f"C5 GNOConv {width}=>{width}, 64x64 grid radius {radius}", l, colmajor(width, 4096), ps, st, reps,
        dict(nodes=4096, edges=int(s.size), kernel_tensor_GB=round(s.size * width * width * 4 / 1e9, 2)))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--traj", type=int, default=8)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--radius", type=float, default=0.05)
    ap.add_argument("--only", default="c3,c4,c5")
    ap.add_argument("--act", default="swish", help="C4 activation (the config says swish; relu isolates the transcendental cost)")
    a = ap.parse_args()
    if "c3" in a.only: c3(a.reps)
    if "c4" in a.only: c4(a.reps, a.traj, a.act)
    if "c5" in a.only: c5(a.reps, a.width, a.radius)

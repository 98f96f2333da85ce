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
    for _ in range(5): fn()
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
    leaves = [x] + grad_leaves(ps)
    def fb():
        for v in leaves:
            v.grad = None                               # gradients are written, not accumulated (no torch adds in the trace)
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
    # the bench line's own workload builder (bench.py: c4_layer), so that profiles and `secondary` run the same graph and data
    import bench as B
    if act != "swish":
        raise SystemExit("bench.c4_layer is the configured (swish) layer")
    l, ps, st, x, n_edges = B.c4_layer(DEV, traj, 0)
    run(f"C4 MPPDEConv h=64, {traj} trajectories x 8192-node periodic mesh (6 neighbours)", l, x, ps, st, reps,
        dict(nodes=8192 * traj, edges=n_edges, trajectories=traj))


def c5(reps, width, radius):
    import bench as B
    l, ps, st, x, n_edges = B.c5_layer(DEV, radius, width)          # synth.grid_radius_graph: the edges of bench.py's `secondary`
    run(f"C5 GNOConv {width}=>{width}, 64x64 grid radius {radius}", l, x, ps, st, reps,
        dict(nodes=4096, edges=n_edges, algorithmic_GFLOP_forward=round(B.c5_fwd_flop(n_edges, width) / 1e9, 2),
             kernel_tensor_GB=round(n_edges * width * width * 4 / 1e9, 2)))


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

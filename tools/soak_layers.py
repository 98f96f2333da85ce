"""Race screen for the round-2 kernels whose lanes hand data to each other through LDS inside a wave or across a workgroup without
a library-level barrier in between -- the one-launch GAT layer (group-private coefficient table, LDS-DMA staged rows) and its
two-launch pullback, the GNO message on the matrix pipe and its pullback, the C4 kernels of DESIGN 5.7 -- and for the persistent
solver: every repetition of
forward + backward must reproduce the first one bit for bit.  usage: python tools/soak_layers.py [repetitions]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import synth as S

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
DEV = "cuda"


def leaves(ps):
    out = []
    for v in ps.values():
        out += leaves(v) if isinstance(v, dict) else [v]
    return out


def soak(name, layer, x, ps, st):
    ps = ng.to_device(ps, DEV)
    for v in leaves(ps):
        v.requires_grad_(True)
    x = x.detach().requires_grad_(True)
    R = None
    first, bad = None, 0
    for r in range(reps):
        for v in leaves(ps) + [x]:
            v.grad = None
        y = layer(x, ps, st)[0]
        if R is None:
            R = torch.randn(y.shape[1], y.shape[0], device=DEV).T
        y.backward(R)
        outs = [y.detach()] + [v.grad for v in [x] + leaves(ps)]
        if first is None:
            first = [o.clone() for o in outs]
        elif not all(torch.equal(a, b) for a, b in zip(first, outs)):
            bad += 1
    torch.cuda.synchronize()
    print(json.dumps({"layer": name, "repetitions": reps, "differing_from_first": bad}), flush=True)
    return bad


torch.manual_seed(0)
_, s, t = S.closest_pairs_graph(16384, 65536, seed=2)
g = ng.GNNGraph(s, t, num_nodes=16384, index_base=0)
bad = 0
gat = ng.GATConv((64, 16), "relu", heads=4, initialgraph=g)
ps, st = ng.setup(3, gat)
bad += soak("C3 GATConv 64 => 4 x 16 (one-launch layer + two-launch pullback)", gat, torch.randn(16384, 64, device=DEV).T, ps, st)
pts, s5, t5 = S.grid_radius_graph(64, 0.1)
g5 = ng.GNNGraph(s5, t5, num_nodes=4096, index_base=0,
                 ndata={"a": S.uniform01(50, 4096).reshape(1, 4096).astype(np.float32), "x": pts.astype(np.float32)})
phi = ng.Chain(ng.Dense(6, 64, "relu"), ng.Dense(64, 128 * 128))
gno = ng.GNOConv((128, 128), phi, "relu", initialgraph=g5)
ps5, st5 = ng.setup(5, gno)
bad += soak("C5 GNOConv 128 => 128, r = 0.1 (message + pullback on the matrix pipe)", gno, torch.randn(4096, 128, device=DEV).T, ps5, st5)
rhs = ng.Chain(ng.GCNConv((64, 64), "relu", initialgraph=g), ng.GCNConv((64, 64), "relu", initialgraph=g))
node = ng.NeuralODE(rhs, solver="tsit5", n_steps=50, dt=0.02)
psn, stn = ng.setup(0, node)
bad += soak("C2 NeuralODE (persistent solver, through the layer API)", node, torch.randn(16384, 64, device=DEV).T, psn, stn)
# MPPDEConv in BASELINE config 4's shape (8 trajectories x 8192-node periodic mesh = 65 536 nodes): the streaming Dense kernels
# (pair, chain, one-launch pullbacks) and the specialised message kernels with their hand-placed LDS reads and waits
n4, traj = 8192, 8
idx = np.arange(n4)
s4 = np.concatenate([idx for k in (-3, -2, -1, 1, 2, 3)]); t4 = np.concatenate([(idx + k) % n4 for k in (-3, -2, -1, 1, 2, 3)])
S4, T4 = np.concatenate([s4 + i * n4 for i in range(traj)]), np.concatenate([t4 + i * n4 for i in range(traj)])
N4 = n4 * traj
g4 = ng.GNNGraph(S4, T4, num_nodes=N4, index_base=0, num_graphs=traj,
                 ndata={"u": torch.rand(1, N4), "x": torch.as_tensor(np.tile(idx / n4, traj)[None, :].astype(np.float32))},
                 gdata={"θ": torch.rand(2, traj)})
mp = ng.MPPDEConv(ng.Chain(ng.Dense(132, 64, "swish"), ng.Dense(64, 64, "swish")), ng.Chain(ng.Dense(130, 64, "swish"), ng.Dense(64, 64)),
                  initialgraph=g4)
psm, stm = ng.setup(4, mp)
bad += soak("C4-shaped MPPDEConv, 8 trajectories (pair / chain Dense launches, one-launch pullbacks, specialised message kernels)", mp,
            torch.randn(N4, 64, device=DEV).T, psm, stm)
sys.exit(1 if bad else 0)

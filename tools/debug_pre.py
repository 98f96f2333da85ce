"""Diagnostic: the pre-scaled (LDS-DMA) NeuralODE plan against the register-staged one, row by row."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
from ngpde_amd.node import _Plan
lib = _lib.load()
N, PAIRS, D, STEPS = 16384, 65536, 64, int(os.environ.get("STEPS", "1"))
TAB = os.environ.get("TAB", "tsit5")
pts, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
h = g.handle((True, None, False))
dev = "cuda"
torch.manual_seed(0)
u0 = torch.randn(N, D, device=dev); w1 = torch.randn(D, D, device=dev) * 0.1; w2 = torch.randn(D, D, device=dev) * 0.1
b1 = torch.randn(D, device=dev) * 0.1; b2 = torch.randn(D, device=dev) * 0.1
seed = torch.randn(N, D, device=dev)
st = torch.cuda.current_stream().cuda_stream; p = _lib.ptr
res = {}
for tag, env in (("pre", None), ("ref", "1")):
    if env: os.environ["NGPDE_NO_PRESCALE"] = env
    plan = _Plan(h, D, int(os.environ.get("ACT", "1")), TAB, STEPS, 0.02, True)
    uT = torch.empty_like(u0); du0 = torch.empty_like(u0)
    dw1 = torch.empty_like(w1); dw2 = torch.empty_like(w1); db1 = torch.empty_like(b1); db2 = torch.empty_like(b1)
    _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(uT), st))
    _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(du0), p(dw1), p(db1), p(dw2), p(db2), st))
    torch.cuda.synchronize()
    res[tag] = dict(uT=uT.cpu().numpy(), du0=du0.cpu().numpy(), dw1=dw1.cpu().numpy(), dw2=dw2.cpu().numpy(), db1=db1.cpu().numpy(), db2=db2.cpu().numpy())
order = g.node_order()
for k in res["pre"]:
    a, b = res["pre"][k], res["ref"][k]
    err = np.abs(a - b)
    print(k, "max err", err.max(), "ref max", np.abs(b).max())
    if a.ndim == 2 and a.shape[0] == N:
        rows = np.nonzero(err.max(axis=1) > 4e-6 * np.abs(b).max())[0]
        pos = np.empty(N, np.int64); pos[order] = np.arange(N)
        print("   bad rows", rows.size, "first", rows[:10], "tile positions", (pos[rows[:10]] // 32), (pos[rows[:10]] % 32))
        deg = np.bincount(t, minlength=N)
        if rows.size: print("   degrees of bad rows", np.bincount(deg[rows]), " all", np.bincount(deg))
# replay determinism of the pre-scaled plan
os.environ.pop("NGPDE_NO_PRESCALE", None)
plan = _Plan(h, D, int(os.environ.get("ACT", "1")), TAB, STEPS, 0.02, True)
runs = []
for rep in range(4):
    uT = torch.empty_like(u0); du0 = torch.empty_like(u0)
    dw1 = torch.empty_like(w1); dw2 = torch.empty_like(w1); db1 = torch.empty_like(b1); db2 = torch.empty_like(b1)
    _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(uT), st))
    _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(du0), p(dw1), p(db1), p(dw2), p(db2), st))
    torch.cuda.synchronize()
    runs.append(dict(uT=uT.cpu().numpy(), du0=du0.cpu().numpy(), dw1=dw1.cpu().numpy(), dw2=dw2.cpu().numpy()))
for k in runs[0]:
    print("replay", k, [float(np.abs(runs[r][k] - runs[0][k]).max()) for r in range(1, 4)], "vs ref", float(np.abs(runs[0][k] - res["ref"][k]).max()))

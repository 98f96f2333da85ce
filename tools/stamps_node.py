"""Diagnostic: per-workgroup phase timestamps of every kernel of one Tsit5 step (libngpde_diag.so, eager)."""
import ctypes as C
import os, sys
os.environ["NGPDE_NODE_EAGER"] = "1"
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
from ngpde_amd.node import _Plan
_lib.LIB_PATH = os.path.join(ROOT, "neuralgraphpde.jl_amd", "libngpde_diag.so")
lib = _lib.load()
lib.ngpde_debug_set_stamps.argtypes = [C.c_void_p, C.c_int32]; lib.ngpde_debug_set_stamps.restype = C.c_int32

N, PAIRS, D, STEPS = 16384, 65536, 64, 2
pts, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
h = g.handle((True, None, False))
plan = _Plan(h, D, 1, "tsit5", STEPS, 0.02, True)
dev = "cuda"
u0 = torch.randn(N, D, device=dev); w1 = torch.randn(D, D, device=dev) * 0.1; w2 = torch.randn(D, D, device=dev) * 0.1
b1 = torch.zeros(D, device=dev); b2 = torch.zeros(D, device=dev)
uT = torch.empty_like(u0); du0 = torch.empty_like(u0); seed = torch.ones_like(u0)
dw1 = torch.empty_like(w1); dw2 = torch.empty_like(w1); db1 = torch.empty_like(b1); db2 = torch.empty_like(b1)
st = torch.cuda.current_stream().cuda_stream; p = _lib.ptr
def solve():
    _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(uT), st))
    _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(du0), p(dw1), p(db1), p(dw2), p(db2), st))
for _ in range(3): solve()
torch.cuda.synchronize()
nb = (N + 31) // 32
nl = 2 * 6 * STEPS + 1 + 2 * 6 * STEPS
stamps = torch.zeros(nl * nb * 16, dtype=torch.int64, device=dev)
_lib.check(lib.ngpde_debug_set_stamps(stamps.data_ptr(), nl))
solve(); torch.cuda.synchronize()
a = stamps.cpu().numpy().reshape(nl, nb, 8, 2)
nf = 2 * 6 * STEPS
def report(name, idx, nph, labels):
    clk = a[idx][:, :, :nph + 1, 0].astype(np.float64); wall = a[idx][:, :, :nph + 1, 1].astype(np.float64)
    used = wall[0, :, 0] > 0                       # paired backward workgroups stamp only n_tiles / 2 slots
    clk, wall = clk[:, used], wall[:, used]
    d = np.diff(clk, axis=2)
    med = np.median(d, axis=(0, 1))
    dur = (wall[:, :, nph].max(axis=1) - wall[:, :, 0].min(axis=1)) * 10.0   # ns, kernel span by wall clock
    print(f"{name}: kernel span (first WG start -> last WG end) median {np.median(dur)/1000:.2f} us over {len(idx)} launches, {int(used.sum())} workgroups")
    for l, m in zip(labels, med): print(f"    {l:26s} {m:8.0f} cycles")
    tot = clk[:, :, nph] - clk[:, :, 0]
    wtot = (wall[:, :, nph] - wall[:, :, 0]) * 10.0
    print(f"    {'WG total':26s} {np.median(tot):8.0f} cycles = {np.median(wtot):.0f} ns;  start skew {np.median((wall[:, :, 0].max(axis=1) - wall[:, :, 0].min(axis=1)) * 10):.0f} ns")
raw = stamps.cpu().numpy().reshape(nl, nb, 16)
sub = raw[list(range(0, nf, 2))][:, :, 10:15].astype(np.float64)
t0 = raw[list(range(0, nf, 2))][:, :, 0].astype(np.float64)
names = ["round-1 data arrived", "halo rows arrived", "LDS written", "barrier passed", "LDS aggregation done"]
print("fwd layer1 halo aggregation sub-phases (cycles since WG start, median / p90 over WGs):")
for k, nme in enumerate(names):
    dlt = sub[:, :, k] - t0
    print(f"    {nme:26s} {np.median(dlt):8.0f} {np.percentile(dlt, 90):8.0f}")
fl = ["sched+W issue+aggregate", "LDS stage+sync", "MFMA+sync", "epilogue"]
report("fwd layer1", list(range(0, nf, 2)), 4, fl)
report("fwd layer2+stage", list(range(1, nf, 2)), 4, fl)
bl = ["sched+W issue+aggregate", "comb+mask+LDS stage+sync", "G MFMA+sync", "G rows out", "dW MFMA+db", "fold+slab store"]
report("bwd layer1", list(range(nf + 1, nl - 1, 2)), 6, bl)
report("bwd stage+layer2", list(range(nf + 2, nl - 1, 2)), 6, bl)

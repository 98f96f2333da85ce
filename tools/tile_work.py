"""Which tiles set the pace of the persistent solver?  Reads the per-workgroup stamp segments that `DUMP=1 python3 tools/stamps_persistent.py`
leaves in gpurun_out/stamps_{fwd,bwd}.npy (diagnostic build) and relates a workgroup's own work per phase (everything but the wait) to its
tile: referenced rows, sum and maximum of the degrees, aggregation rounds of its slowest wave, and the XCD it runs on.  Finding of round 5
(DESIGN 5.2): the phase period is the cycle through two neighbouring tiles -- the mean of their own work plus one hand-off; own work spreads 5.1 k - 6.8 k cycles (5 - 95 %) in
the forward with correlations of only 0.2 - 0.3 to any tile property (+ ~150 cycles per aggregation round); the workgroups of XCD 2 are
15 % slower in the forward launch on every box measured, with tiles like everybody else's."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
N, PAIRS = 16384, 65536
_, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
h = g.handle((True, None, False))
order = np.asarray(g.node_order())
deg = np.bincount(t, minlength=N)
nt = N // 32
lib = _lib.load()
ptr, nbytes = C.c_void_p(), C.c_size_t()
_lib.check(lib.ngpde_graph_array(h.ptr, 0, 8, C.byref(ptr), C.byref(nbytes)))     # by-target tile_info
info = torch.empty(nbytes.value // 4, dtype=torch.int32, device="cuda")
import ctypes
torch.cuda.synchronize()
# copy device -> tensor through hipMemcpy via torch: wrap the pointer
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(ctypes.c_void_p(info.data_ptr()), ptr, ctypes.c_size_t(nbytes.value), 3)
hc = info.cpu().numpy().reshape(-1, 2)[:, 0]
td = deg[order].reshape(nt, 32)
tile_max, tile_sum = td.max(1), td.sum(1)
wave_max = td.reshape(nt, 8, 4).max(2)
rounds = np.ceil(wave_max / 4).max(1)
for which in ("fwd", "bwd"):
    a = np.load(f"gpurun_out/stamps_{which}.npy")
    b = np.arange(nt)
    tile_of_b = (b % 8) * (nt // 8) + b // 8
    own = a[:, 1:].sum(1)
    agg = a[:, 2]
    for name, x in (("halo count", hc[tile_of_b]), ("sum of degrees", tile_sum[tile_of_b]), ("max degree", tile_max[tile_of_b]), ("rounds of the slowest wave", rounds[tile_of_b])):
        print(which, f"corr(own work, {name}) = {np.corrcoef(own, x)[0,1]:.2f}   corr(aggregation, {name}) = {np.corrcoef(agg, x)[0,1]:.2f}")
    for x in range(8):
        sel = b % 8 == x
        print(f"  xcd {x}: halo {hc[tile_of_b][sel].mean():.1f} sum deg {tile_sum[tile_of_b][sel].mean():.0f} max deg {tile_max[tile_of_b][sel].mean():.1f} | own {own[sel].mean():.0f} agg {agg[sel].mean():.0f}")
for which in ("fwd", "bwd"):
    a = np.load(f"gpurun_out/stamps_{which}.npy")
    b = np.arange(nt)
    tile_of_b = (b % 8) * (nt // 8) + b // 8
    own = a[:, 1:].sum(1); agg = a[:, 2]
    r = rounds[tile_of_b]
    for k in np.unique(r):
        sel = (r == k) & (b % 8 != 2)
        print(which, f"rounds {int(k)}: {sel.sum()} tiles, aggregation {agg[sel].mean():.0f}, own {own[sel].mean():.0f}, max own {own[sel].max():.0f}")
    ts = tile_sum[tile_of_b]
    for lo, hi in ((0, 230), (230, 260), (260, 290), (290, 400)):
        sel = (ts >= lo) & (ts < hi) & (b % 8 != 2)
        print(which, f"sum of degrees {lo}-{hi}: {sel.sum()} tiles, aggregation {agg[sel].mean():.0f}, own {own[sel].mean():.0f}")

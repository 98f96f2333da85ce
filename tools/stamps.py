"""Diagnostic: per-workgroup phase timestamps of the fused forward kernel (libngpde_diag.so)."""
import ctypes as C
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
_lib.LIB_PATH = os.path.join(ROOT, "neuralgraphpde.jl_amd", "libngpde_diag.so")
lib = _lib.load()
lib.ngpde_debug_set_stamps.argtypes = [C.c_void_p]; lib.ngpde_debug_set_stamps.restype = C.c_int32

N, PAIRS, D = 16384, 65536, 64
pts, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
h = g.handle((True, None, False))
x = torch.randn(N, D, device="cuda"); w = torch.randn(D, D, device="cuda") * 0.1; b = torch.zeros(D, device="cuda")
y = torch.empty_like(x); agg = torch.empty_like(x)
ws = torch.empty(1024, dtype=torch.uint8, device="cuda")
nb = (N + 31) // 32
stamps = torch.zeros(nb * 16, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
p = _lib.ptr
fwd = lambda: _lib.check(lib.ngpde_gcn_forward(h.ptr, D, D, 1, p(x), p(w), p(b), p(y), p(agg), None, p(ws), ws.numel(), st))
for _ in range(10): fwd()
torch.cuda.synchronize()
_lib.check(lib.ngpde_debug_set_stamps(stamps.data_ptr()))
fwd(); torch.cuda.synchronize()
a = stamps.cpu().numpy().reshape(nb, 8, 2)
clk, wall = a[:, :5, 0], a[:, :5, 1]
names = ["W-issue+aggregate", "LDS stage+sync", "MFMA+sync", "epilogue"]
d = np.diff(clk, axis=1)
print("phase cycles (shader clock) median / p90 / max over", nb, "workgroups")
for k, nme in enumerate(names):
    print(f"  {nme:20s} {np.median(d[:, k]):8.0f} {np.percentile(d[:, k], 90):8.0f} {d[:, k].max():8.0f}")
tot = clk[:, 4] - clk[:, 0]
print(f"  {'WG total':20s} {np.median(tot):8.0f} {np.percentile(tot, 90):8.0f} {tot.max():8.0f}")
w0 = wall[:, 0].min()
print("wall (100 MHz ticks = 10 ns): first WG start 0; last WG start", wall[:, 0].max() - w0, "; last WG end", wall[:, 4].max() - w0)
starts = np.sort(wall[:, 0] - w0)
print("WG start quantiles (x10ns):", [int(starts[int(q * (nb - 1))]) for q in (0, .25, .5, .75, .9, 1)])
ends = np.sort(wall[:, 4] - w0)
print("WG end   quantiles (x10ns):", [int(ends[int(q * (nb - 1))]) for q in (0, .25, .5, .75, .9, 1)])
print("cycles per 10ns tick ~", np.median(tot / np.maximum(wall[:, 4] - wall[:, 0], 1)))

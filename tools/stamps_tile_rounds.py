"""Where does a TURN of the pipelined tile-round forward kernel go (node_fwd_persistentKP_kernel)?  Needs the diagnostic library
(make -C neuralgraphpde.jl_amd/csrc diag-persistent).  Stamps of thread 0 of every workgroup at 7 points of the first TURNS turns."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S

_lib.LIB_PATH = os.path.join(ROOT, "neuralgraphpde.jl_amd", "libngpde_diag.so")
from ngpde_amd.node import _Plan

N = int(os.environ.get("N", 65536))
D, STEPS, TURNS = 64, 10, 400
dev = "cuda:0"
_, s, t = S.closest_pairs_graph(N, 4 * N, seed=4)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
lib, p = _lib.load(), _lib.ptr
lib.ngpde_debug_set_persistent_stamps.argtypes = [C.c_void_p, C.c_int32]
dv = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev)
u0 = dv(S.normal(1000, D * N).reshape(N, D))
w1, w2 = dv(S.glorot_uniform(11, D, D).T), dv(S.glorot_uniform(12, D, D).T)
b1, b2 = dv(np.zeros(D)), dv(np.zeros(D))
stream = torch.cuda.current_stream().cuda_stream
plan = _Plan(g.handle((True, None, False)), D, _lib.ACT["relu"], "tsit5", STEPS, 1.0 / 50, True)
assert "tile_rounds" in plan.flags(), plan.flags()
uT = torch.empty_like(u0)
NW = 512
buf = torch.zeros(NW * TURNS * 8, dtype=torch.int64, device=dev)
for rep in range(3):
    lib.ngpde_debug_set_persistent_stamps(p(buf) if rep == 2 else None, TURNS)
    _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(uT), stream))
    torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(NW, TURNS, 8).astype(np.float64)
names = ["T0: vmcnt(0) + barrier (gather landed, stores drained)", "publish + blocking path if the gather was not ahead", "poll issue + aggregation + barrier",
         "next turn's DMA issue", "product + barrier", "epilogue, stores, next state loads"]
sel = st[:, 16:TURNS - 1, :]
nxt = st[:, 17:TURNS, 0]
d = np.diff(np.concatenate([sel[:, :, :7], nxt[:, :, None]], axis=2), axis=2)
print(f"nodes {N}, flags {sorted(plan.flags())}")
for k, nm in enumerate(names + ["(loop overhead to the next turn)"]):
    print(f"   {nm}: mean {d[:, :, k].mean():.0f}  p10 {np.percentile(d[:, :, k], 10):.0f}  p90 {np.percentile(d[:, :, k], 90):.0f}")
print(f"   turn start-to-start: {(nxt - sel[:, :, 0]).mean():.0f} cycles")

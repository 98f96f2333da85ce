"""Diagnostic (libngpde_diag.so, `make -C neuralgraphpde.jl_amd/csrc diag`): phase timestamps of the pipelined 64-wide
message kernel (edge_mlp64.hip) on the C4 shard, one steady-state tile per persistent workgroup."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "neuralgraphpde.jl_amd", "libngpde_diag.so")
lib = _lib.load()
lib.ngpde_debug_set_edge64_stamps.argtypes = [C.c_void_p]; lib.ngpde_debug_set_edge64_stamps.restype = C.c_int32
DEV = "cuda:0"
n, h, traj = 8192, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 64
idx = np.arange(n)
s = np.concatenate([idx for k in (-3, -2, -1, 1, 2, 3)]); t = np.concatenate([(idx + k) % n for k in (-3, -2, -1, 1, 2, 3)])
S_, T_ = np.concatenate([s + i * n for i in range(traj)]), np.concatenate([t + i * n for i in range(traj)])
N = n * traj
g = ng.GNNGraph(S_, T_, num_nodes=N, index_base=0, num_graphs=traj,
                ndata={"u": torch.rand(1, N), "x": torch.as_tensor(np.tile(idx / n, traj)[None, :].astype(np.float32))},
                gdata={"θ": torch.rand(2, traj)})
act = sys.argv[2] if len(sys.argv) > 2 else "swish"
l = ng.MPPDEConv(ng.Chain(ng.Dense(132, 64, act), ng.Dense(64, 64, act)), ng.Chain(ng.Dense(130, 64, act), ng.Dense(64, 64)), initialgraph=g)
ps, st = ng.setup(4, l)
ps = ng.to_device(ps, DEV)
x = torch.randn(N, h, device=DEV).T
nb = 512
stamps = torch.zeros(nb * 16, dtype=torch.int64, device=DEV)
with torch.no_grad():
    for _ in range(3): l(x, ps, st)
    torch.cuda.synchronize()
    lib.ngpde_debug_set_edge64_stamps(stamps.data_ptr())
    l(x, ps, st); torch.cuda.synchronize()
a = stamps.cpu().numpy().reshape(nb, 16)
a = a[a[:, 13] > 0]
names = ["stage + prefix + edge words (3 syncs)", "a1 of slice 0", "products 0 | a1 of slice 1", "issue next tile's row loads",
         "steady block (it = 1)", "barrier", "reduce", "barrier", "steady block (it = 2) + reduce + 2 barriers", "(loop exit)",
         "messages of the last slice", "barrier + reduce + barrier", "output store"]
d = np.diff(a[:, :14], axis=1)
print(f"{a.shape[0]} workgroups; phase cycles (shader clock) median / p90")
for k, nm in enumerate(names):
    print(f"  {nm:46s} {np.median(d[:, k]):8.0f} {np.percentile(d[:, k], 90):8.0f}")
tot = a[:, 13] - a[:, 0]
wall = a[:, 15] - a[:, 14]
print(f"  {'tile total':46s} {np.median(tot):8.0f} {np.percentile(tot, 90):8.0f}")
print("shader cycles per 10 ns wall tick:", np.median(tot / np.maximum(wall, 1)), " => clock GHz ~", np.median(tot / np.maximum(wall, 1)) / 10)

"""Where does a PHASE (one right-hand-side evaluation) of the device-resident VMH solver go?  Needs the diagnostic library
(make -C neuralgraphpde.jl_amd/csrc diag-vmh).  Shader-clock stamps of thread 0 of every workgroup at 7 points of every phase,
forward launch and adjoint launch.  env: N (3000), K (6), STEPS (20)"""
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
dev = "cuda:0"
nv, kv, steps = int(os.environ.get("N", 3000)), int(os.environ.get("K", 6)), int(os.environ.get("STEPS", 20))
pts = torch.as_tensor(S.uniform01(41, 2 * nv).reshape(2, nv).astype(np.float32), device=dev)
gv = ng.GNNGraph(ng.knn_graph(pts, kv), ndata={"x": pts})
phi = ng.Chain(ng.Dense(4, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 40))
gam = ng.Chain(ng.Dense(41, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 1))
u0 = torch.as_tensor(S.normal(42, nv).reshape(1, nv).astype(np.float32), device=dev)
node = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=gv), solver="tsit5", n_steps=steps, dt=0.2 / steps, capture=False)
ps, st = ng.setup(4, node)
ps = ng.to_device(ps, dev)


def leaves(t):
    for v in t.values():
        if isinstance(v, dict):
            yield from leaves(v)
        else:
            yield v


for v in leaves(ps):
    v.requires_grad_(True)
lib = _lib.load()
lib.ngpde_debug_set_vmh_stamps.argtypes = [C.c_void_p, C.c_int32]
PH, NW = 6 * steps, 512
buf = torch.zeros(NW * PH * 8, dtype=torch.int64, device=dev)
u = u0.clone().requires_grad_(True)
for rep in range(3):
    lib.ngpde_debug_set_vmh_stamps(_lib.ptr(buf) if rep == 2 else None, PH)
    uT, _ = node(u, ps, st)
    torch.cuda.synchronize()
    fw = buf.cpu().numpy().reshape(NW, PH, 8).astype(np.float64)
    buf.zero_()
    uT.sum().backward()
    torch.cuda.synchronize()
    bw = buf.cpu().numpy().reshape(NW, PH, 8).astype(np.float64)
plans = [p for pool in node._plans.values() for p in pool]
print("plans", [sorted(p.flags()) for p in plans])


def report(title, st, names):
    used = st[:, 0, 0] > 0
    st = st[used]
    sel, nxt = st[:, 8:PH - 1, :7], st[:, 9:PH, 0]
    d = np.diff(np.concatenate([sel, nxt[:, :, None]], axis=2), axis=2)
    print(f"{title}: {int(used.sum())} workgroups, phase start-to-start {(nxt - sel[:, :, 0]).mean():.0f} cycles (100 MHz... no: shader clock)")
    for k, nm in enumerate(names):
        print(f"   {nm}: mean {d[:, :, k].mean():.0f}  p10 {np.percentile(d[:, :, k], 10):.0f}  p90 {np.percentile(d[:, :, k], 90):.0f}")


report("forward", fw, ["wait for the neighbours' flags", "halo values + barrier", "message MLP of the wave's 16 edges", "staging + per-target sums",
                       "node MLP (4 layers, a barrier each)", "stage update + drain + flag", "tape rows issued, loop overhead"])
if (fw[:, 8:PH - 1, 7] > 0).any():      # the forward's staging step, split at its first barrier (stamp 7)
    usedf = fw[:, 0, 0] > 0
    f = fw[usedf][:, 8:PH - 1, :]
    print(f"   staging, split: message MLP end of wave 0 -> all waves staged {np.mean(f[:, :, 7] - f[:, :, 3]):.0f}, "
          f"-> sums done {np.mean(f[:, :, 4] - f[:, :, 7]):.0f}")
report("adjoint", bw, ["K-bar + node MLP backwards", "message MLP backwards of the wave's 16 edges", "barrier (the slowest wave)", "own-row sums + drain + flag",
                       "dz rows issued", "wait for the neighbours' flags", "by-source gather + stage adjoint"])

if int(os.environ.get("N", 3000)) > 4096:      # tile rounds: a turn of the adjoint = tables, second half of the phase before, first half of this one
    used = bw[:, 0, 0] > 0
    b = bw[used][:, 8:PH - 1, :]
    seq = [7, 5, 6, 0, 1, 2, 3, 4]
    names = ["tables, by-source positions, state, tape rows asked for", "wait for the phase before (published a sweep ago)", "by-source gather + stage adjoint",
             "K-bar + node MLP backwards", "message MLP backwards (both rounds)", "barrier", "own-row sums + drain + flag"]
    print("adjoint, last turn of a sweep (tile rounds):")
    for k, nm in enumerate(names):
        dd = b[:, :, seq[k + 1]] - b[:, :, seq[k]]
        print(f"   {nm}: mean {dd.mean():.0f}  p10 {np.percentile(dd, 10):.0f}  p90 {np.percentile(dd, 90):.0f}")

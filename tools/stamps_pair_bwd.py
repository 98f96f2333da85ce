"""Diagnostic (libngpde_diag.so, `make diag-dense`): phase timestamps of dense_pair64_bwd_kernel (the pullback of C4's P and Q: dP, dQ, h
-> dh, dW, db), one steady-state tile per persistent workgroup.  DESIGN 5.4."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng  # noqa: F401
from ngpde_amd import _lib
sys.path.insert(0, os.path.join(ROOT, "tests"))
import composed as F      # the primitives' autograd wrappers (tests/composed.py)
_lib.LIB_PATH = os.path.join(ROOT, "neuralgraphpde.jl_amd", "libngpde_diag.so")
lib = _lib.load()
lib.ngpde_debug_set_pair_bwd_stamps.argtypes = [C.c_void_p]; lib.ngpde_debug_set_pair_bwd_stamps.restype = C.c_int32
DEV = "cuda:0"
n = 524288
h = torch.randn(n, 64, device=DEV, requires_grad=True); d = torch.randn(n, 2, device=DEV); th = torch.randn(64, 2, device=DEV)
wp = (torch.randn(68, 64, device=DEV) * 0.1).requires_grad_(True); wq = (torch.randn(66, 64, device=DEV) * 0.1).requires_grad_(True)
bp = torch.randn(64, device=DEV, requires_grad=True)
gp, gq = torch.randn(n, 64, device=DEV), torch.randn(n, 64, device=DEV)
nb = 512
stamps = torch.zeros(nb * 16, dtype=torch.int64, device=DEV)


def once():
    ya, yb = F.dense_pair([h, d, th], wp, bp, 0, [h, d], wq, None, 0, row_divs_a=[1, 1, n // 64], n=n)
    torch.autograd.backward([ya, yb], [gp, gq])
    h.grad = wp.grad = wq.grad = bp.grad = None


for _ in range(3):
    once()
torch.cuda.synchronize()
lib.ngpde_debug_set_pair_bwd_stamps(stamps.data_ptr())
once(); torch.cuda.synchronize()
a = stamps.cpu().numpy().reshape(nb, 16)
a = a[a[:, 9] > 0]
names = ["top barrier", "dy rows -> LDS + barrier", "issue the next tile's loads", "X DMA issue + products (256 MFMAs per wave)",
         "collect the next tile's loads (vmcnt)", "addend loads issued + barrier", "stage dX + barrier", "addend wait", "dX stores issued"]
dd = np.diff(a[:, :10], axis=1)
print(f"{a.shape[0]} workgroups; phase cycles (s_memtime) median / p90")
for k, nm in enumerate(names):
    print(f"  {nm:48s} {np.median(dd[:, k]):8.0f} {np.percentile(dd[:, k], 90):8.0f}")
print(f"  {'tile total':48s} {np.median(a[:, 9] - a[:, 0]):8.0f}")
print(f"  inside the products phase (wave 0): X DMA issue {np.median(a[:, 12] - a[:, 3]):.0f}, dW (160 MFMAs) {np.median(a[:, 13] - a[:, 12]):.0f}, "
      f"dX (128 MFMAs) {np.median(a[:, 4] - a[:, 13]):.0f}")

# who shares a CU: HW_ID bits 8..11 = CU, 13..15 = SE (gfx9 layout), XCC_ID bits 0..3
hw, xcc = a[:, 10].astype(np.int64), a[:, 11].astype(np.int64) & 15
cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (xcc << 8)
blocks = np.arange(nb)[stamps.cpu().numpy().reshape(nb, 16)[:, 9] > 0]
groups = {}
for b_, c_ in zip(blocks, cu):
    groups.setdefault(int(c_), []).append(int(b_))
sizes = np.bincount([len(v) for v in groups.values()])
print("workgroups per (XCD, SE, CU):", {k: int(v) for k, v in enumerate(sizes) if v})
pairs = [v for v in groups.values() if len(v) == 2]
print("first pairs sharing a CU (block ids):", pairs[:8])
delta = [b2 - b1 for b1, b2 in (sorted(v) for v in pairs)]
print("block-id distance within a pair: ", dict(zip(*np.unique(delta, return_counts=True))))
row = {int(b_): k for k, b_ in enumerate(blocks)}
off = []
for v in pairs:
    t0, t1 = a[row[v[0]], 3], a[row[v[1]], 3]      # start of the products phase of the stamped tile
    off.append(abs(int(t0) - int(t1)))
off = np.array(off)
print(f"|offset| between the products-phase starts of a CU's two workgroups: median {np.median(off):.0f}, p10 {np.percentile(off, 10):.0f}, p90 {np.percentile(off, 90):.0f} cycles"
      f" (tile {np.median(a[:, 9] - a[:, 0]):.0f})")

"""Microbenchmark of the segmented Dense primitive (ngpde_dense_forward / _backward) on row-major activations.
usage: python tools/bench_dense.py [--n 524288] ; NGPDE_DENSE_NARROW=1 forces the 64-row tile kernels."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import functional as F

DEV = "cuda:0"


def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def case(n, widths, dout, act):
    blocks = [torch.randn(n, w, device=DEV, requires_grad=True) for w in widths]
    din = sum(widths)
    wt = torch.randn(din, dout, device=DEV, requires_grad=True)
    b = torch.randn(dout, device=DEV, requires_grad=True)
    R = torch.randn(n, dout, device=DEV)
    with torch.no_grad():
        f_inf = timeit(lambda: F.dense(blocks, wt, b, act))
    f_tr = timeit(lambda: F.dense(blocks, wt, b, act))
    def fb():
        F.dense(blocks, wt, b, act).backward(R)
    t_fb = timeit(fb)
    gflop = 2.0 * n * din * dout / 1e9
    mb = 4.0 * n * (din + dout) / 1e6
    print(json.dumps(dict(n=n, widths=widths, dout=dout, act=act, ms_fwd_infer=round(f_inf, 3), ms_fwd_train=round(f_tr, 3),
                          ms_fwd_bwd=round(t_fb, 3), fwd_TFLOPs=round(gflop / f_inf, 1), fwd_GBs=round(mb / f_inf, 0))), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=524288)
    a = ap.parse_args()
    case(a.n, [64], 64, 0)
    case(a.n, [64], 64, 4)
    case(a.n, [64, 1, 1, 2], 64, 4)
    case(a.n, [64, 64, 2], 64, 4)
    case(a.n, [128], 128, 1)
    case(4096, [128], 8192, 0)

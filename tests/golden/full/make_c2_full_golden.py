"""Float64 golden vectors of the FULL bench workload (BASELINE config 2: 16 384 nodes, 65 536 closest pairs, 64-d,
Chain(GCNConv(64 => 64, relu) x 2), Tsit5 x 50, dt = 1/50, loss = sum(u(T))), computed by the numpy restatement
oracle/ngpde_oracle.py (gcn2_node_loss_and_grads: /root/reference/src/layers.jl:200-239 under the fixed-step solver of
docs/src/tutorials/graph_node.md:44-66).  Takes ~4-6 minutes of one host core, which is why it is a file of its own and its
output is committed: tests/test_configs_gpu.py compares the HIP path with it at SURVEY.md 8(d)'s tolerances.

Inputs are the bench's own (synth.py, splitmix64 streams), rounded to float32 first -- the values the device sees -- and then
carried in float64.  Stored: dW1, db1, dW2, db2 in full; 256 sampled columns (nodes) of u(T) and du0 with their indices; the
l2 norms of the full fields; a float64 checksum of every input so that a test can tell "the generator's inputs changed" from
"the kernels changed".

usage:  python tests/golden/full/make_c2_full_golden.py        (rewrites tests/golden/full/c2_full_tsit5x50.npz)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
import importlib.util  # noqa: E402

from oracle import ngpde_oracle as O  # noqa: E402

# synth.py is numpy-only; load it by path so that generating the vectors needs neither torch nor the HIP library
_spec = importlib.util.spec_from_file_location("ngpde_synth", os.path.join(ROOT, "neuralgraphpde.jl_amd", "synth.py"))
S = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(S)

N, PAIRS, D, NSTEPS = 16384, 65536, 64, 50


def inputs():
    """the bench workload's inputs (the same calls as tests/test_configs_gpu.py::c2_inputs), float32-rounded"""
    _, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
    r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
    params = [dict(weight=r32(S.glorot_uniform(11 + k, D, D)), bias=np.zeros((D, 1))) for k in range(2)]
    u0 = r32(S.normal(1000, D * N).reshape(N, D).T)
    return s, t, params, u0


def sample_columns():
    # 256 nodes spread over the whole index range by a fixed stride pattern (RNG-free)
    return (np.arange(256, dtype=np.int64) * 6151 + 17) % N


def main():
    s, t, params, u0 = inputs()
    t0 = time.time()
    uT, du0, acc = O.gcn2_node_loss_and_grads(params, O.Graph(s, t, num_nodes=N, index_base=0), u0, O.TABLEAUS["tsit5"],
                                              1.0 / NSTEPS, NSTEPS, "relu")
    print(f"float64 solve + adjoint: {time.time() - t0:.0f} s", flush=True)
    cols = sample_columns()
    out = dict(cols=cols, uT_cols=uT[:, cols], du0_cols=du0[:, cols], uT_norm=np.linalg.norm(uT), du0_norm=np.linalg.norm(du0),
               uT_absmax=np.abs(uT).max(), du0_absmax=np.abs(du0).max(),
               dW1=acc[0]["weight"], db1=acc[0]["bias"], dW2=acc[1]["weight"], db2=acc[1]["bias"],
               in_checksum=np.array([u0.sum(), params[0]["weight"].sum(), params[1]["weight"].sum(), float(s.sum()), float(t.sum())]))
    np.savez_compressed(os.path.join(HERE, "c2_full_tsit5x50.npz"), **out)
    print("wrote c2_full_tsit5x50.npz", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()

"""Why did (or did not) a graph get the persistent solver's hub geometry?  Creates the C1-shaped plan at d = 64 and prints the plan's
flags and the library's last message (node_persistent_setup names the cap a tile exceeded)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
from ngpde_amd.node import _Plan

N, PAIRS = int(os.environ.get("N", 2708)), int(os.environ.get("PAIRS", 5278))
s, t = S.preferential_pairs_graph(N, PAIRS, seed=1)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
h = g.handle((True, None, False))
for d in (64, 32):
    plan = _Plan(h, d, _lib.ACT["relu"], "euler", 10, 0.1, True)
    print("d =", d, "flags:", sorted(plan.flags()), "| last message:", _lib.load().ngpde_last_error().decode())

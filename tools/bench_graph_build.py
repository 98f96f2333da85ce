"""Time of ngpde_graph_create + ngpde_graph_set_gcn_norm (derived-graph handle: CSR by target / source, locality
schedule, halo lists) for the C2 and C4 graph shapes -- the per-minibatch cost when `updategraph` swaps graphs."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S


def timed(make, reps=3):
    ts = []
    for _ in range(reps):
        g = make()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.handle((True, None, False))        # GCN-normalised handle (self loops), as GCNConv asks for
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        g.handle()                           # plain handle, as the edge-function layers ask for
        torch.cuda.synchronize()
        ts.append((t1 - t0, time.perf_counter() - t1))
    return [round(min(t[i] for t in ts) * 1e3, 2) for i in (0, 1)]


if __name__ == "__main__":
    torch.zeros(1, device="cuda:0")
    _, s, t = S.closest_pairs_graph(16384, 65536, seed=2)
    a = timed(lambda: ng.GNNGraph(s, t, num_nodes=16384, index_base=0))
    print(json.dumps(dict(graph="C2 16384 nodes / 131072 edges", ms_gcn_handle=a[0], ms_plain_handle=a[1])), flush=True)
    for traj in (8, 64):
        n = 8192
        idx = np.arange(n)
        s1 = np.concatenate([idx for k in (-3, -2, -1, 1, 2, 3)])
        t1 = np.concatenate([(idx + k) % n for k in (-3, -2, -1, 1, 2, 3)])
        S_, T_ = np.concatenate([s1 + i * n for i in range(traj)]), np.concatenate([t1 + i * n for i in range(traj)])
        a = timed(lambda: ng.GNNGraph(S_, T_, num_nodes=n * traj, index_base=0, num_graphs=traj))
        print(json.dumps(dict(graph=f"C4 {traj} x 8192-node mesh: {n * traj} nodes / {S_.size} edges", ms_gcn_handle=a[0],
                              ms_plain_handle=a[1])), flush=True)
    if os.environ.get("NGPDE_HOST_GRAPH_BUILD") == "1":
        sys.exit(0)

    # graphs built from point clouds on the device (radius_graph / knn_graph) and their first handle: BFS schedule
    # (host traversal of the new structure) against the space-filling-curve schedule (no host traversal)
    def search(make, reps=5):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g = make()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            g.handle((True, None, False))
            torch.cuda.synchronize()
            ts.append((t1 - t0, time.perf_counter() - t1))
        return g, [round(min(t[i] for t in ts) * 1e3, 2) for i in (0, 1)]

    pts = torch.as_tensor(np.stack([S.uniform01(2, 16384), S.uniform01(3, 16384)]).astype(np.float32), device="cuda:0")
    for loc in ("bfs", "spatial"):
        g, a = search(lambda: ng.radius_graph(pts, 0.01247, locality=loc))
        print(json.dumps(dict(graph=f"radius_graph 16384 points r=0.01247 -> {g.num_edges} edges, locality={loc}",
                              ms_search=a[0], ms_first_gcn_handle=a[1])), flush=True)
    g, a = search(lambda: ng.knn_graph(pts, 8, locality="spatial"))
    print(json.dumps(dict(graph=f"knn_graph 16384 points k=8 -> {g.num_edges} edges, locality=spatial", ms_search=a[0],
                          ms_first_gcn_handle=a[1])), flush=True)
    xs = (np.arange(64, dtype=np.float32) + 0.5) / 64
    grid = torch.as_tensor(np.stack([a.reshape(-1) for a in np.meshgrid(xs, xs, indexing="ij")]), device="cuda:0")
    for r in (0.05, 0.1):
        g, a = search(lambda: ng.radius_graph(grid, r, locality="spatial"))
        print(json.dumps(dict(graph=f"C5 radius_graph 64x64 grid r={r} -> {g.num_edges} edges, locality=spatial", ms_search=a[0],
                              ms_first_gcn_handle=a[1])), flush=True)
    big = torch.as_tensor(np.stack([S.uniform01(7, 1 << 20), S.uniform01(8, 1 << 20)]).astype(np.float32), device="cuda:0")
    g, a = search(lambda: ng.radius_graph(big, 0.00156, locality="spatial"), reps=3)
    print(json.dumps(dict(graph=f"radius_graph 1048576 points r=0.00156 -> {g.num_edges} edges, locality=spatial", ms_search=a[0],
                          ms_first_gcn_handle=a[1])), flush=True)

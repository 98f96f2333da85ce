/*
 * ngpde.h -- C ABI of libngpde_hip.so: the MI355X (gfx950) implementation of the message-passing
 * hot path of NeuralGraphPDE.jl (edge gather -> per-edge function -> node scatter-reduce, plus the
 * dense node-feature x weight contraction), i.e. what a Julia `ccall` shim replacing the bodies of
 * /root/reference/src/layers.jl would bind.  See INTEGRATION.md for the reference-side binding.
 *
 * Conventions
 *   - Every function returns an int32 status (0 = NGPDE_OK, < 0 = error); the message of the last
 *     error on the calling thread is returned by ngpde_last_error().  No C++ exception or abort()
 *     crosses this boundary.  The reference raises AssertionError / DimensionMismatch at the same
 *     places (src/layers.jl:204,207,216 and GNNGraph's check_num_nodes/edges).
 *   - All tensor arguments are raw DEVICE pointers to float32 unless a parameter is documented as
 *     "host".  A Julia (D x N) column-major matrix is passed as-is: it is the row-major [N][D]
 *     array the kernels use (node n = D contiguous floats).  A weight (out x in) column-major is
 *     the row-major [in][out] array.
 *   - `stream` is a hipStream_t (NULL = the default stream).  Forward/backward calls only enqueue
 *     work; they never allocate, free or synchronise (graph-capture safe).  Handle creation is
 *     synchronous.
 *   - The library keeps no hidden parameter state: parameters and inputs are passed on every call,
 *     as in Lux's `y, st = layer(x, ps, st)` (src/layers.jl:200, :94, :312, :390, :509).  The only
 *     cached object is the derived-graph handle, whose life is tied to the GNNGraph in `st.graph`.
 */
#ifndef NGPDE_H
#define NGPDE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGPDE_VERSION_STRING "0.1.0"

typedef struct ngpde_graph ngpde_graph_t;     /* derived graph: CSR by target + CSR by source  */
typedef struct ngpde_node ngpde_node_t;       /* fixed-step neural-ODE plan over 2 x GCNConv   */
typedef struct ngpde_node_gat ngpde_node_gat_t; /* fixed-step neural-ODE plan over one GAT-style layer */
typedef struct ngpde_node_vmh ngpde_node_vmh_t; /* fixed-step neural-ODE plan over VMHConv(phi, gamma) */
typedef void *ngpde_stream_t;                 /* hipStream_t                                   */

typedef enum {
  NGPDE_OK = 0,
  NGPDE_ERR_INVALID_ARGUMENT = -1,   /* Julia: ArgumentError / AssertionError */
  NGPDE_ERR_DIMENSION_MISMATCH = -2, /* Julia: DimensionMismatch              */
  NGPDE_ERR_HIP = -3,                /* a HIP runtime call failed             */
  NGPDE_ERR_UNSUPPORTED = -4,        /* shape/option outside what is built    */
  NGPDE_ERR_WORKSPACE = -5,          /* caller workspace too small            */
  NGPDE_ERR_STATE = -6               /* call order violated (e.g. norm not set, backward before forward) */
} ngpde_status_t;

/* NNlib activation names used by the reference's layers (src/layers.jl:180, Lux Dense).  exp / log / reciprocal are the
 * hardware v_exp_f32 / v_log_f32 / v_rcp_f32 (1 ulp): tanh, sigmoid, swish, gelu (tanh form, as NNlib 0.8), elu and softplus
 * are within ~2e-7 absolute of the libm forms, two orders below the parity tolerance of 1e-4. */
typedef enum {
  NGPDE_ACT_IDENTITY = 0, NGPDE_ACT_RELU = 1, NGPDE_ACT_TANH = 2, NGPDE_ACT_SIGMOID = 3,
  NGPDE_ACT_SWISH = 4, NGPDE_ACT_GELU = 5, NGPDE_ACT_LEAKYRELU = 6, NGPDE_ACT_ELU = 7,
  NGPDE_ACT_SOFTPLUS = 8
} ngpde_act_t;

/* `aggr` of propagate(...): +, mean, max, min, * (src/layers.jl:49,257,348,441; fields :90,303,384,496; default mean).
 * The product of an empty neighbourhood is 1 (NNlib scatter(*) starts from the neutral element); its pullback gives every
 * entry the product of the others.  The fused message kernel (ngpde_edge_mlp_forward) takes the first four. */
typedef enum { NGPDE_AGGR_SUM = 0, NGPDE_AGGR_MEAN = 1, NGPDE_AGGR_MAX = 2, NGPDE_AGGR_MIN = 3, NGPDE_AGGR_MUL = 4 } ngpde_aggr_t;

typedef enum { NGPDE_TABLEAU_EULER = 0, NGPDE_TABLEAU_TSIT5 = 1 } ngpde_tableau_t;

const char *ngpde_version(void);
const char *ngpde_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Derived-graph handle.  Replaces what the reference recomputes on EVERY layer call:
 * add_self_loops (src/layers.jl:211), degree (:224) and, inside propagate [GraphNeuralNetworks.jl],
 * the COO -> sparse-matrix build / gather+scatter index walk (:228-232, :111, :326, :416, :534).
 *   s, t        host, int64, length n_edges, the COO vectors of the GNNGraph (edge e: s[e] -> t[e])
 *   index_base  1 for the vectors a Julia GNNGraph holds, 0 for 0-based callers
 *   n_graphs    number of graphs of a batched (block-diagonal) GNNGraph (test/runtests.jl:89-102)
 * The handle is immutable after ngpde_graph_set_gcn_norm and may be shared across streams.
 * ---------------------------------------------------------------------------------------------- */
int32_t ngpde_graph_create(int64_t n_nodes, int64_t n_edges, const int64_t *s, const int64_t *t,
                           int32_t index_base, int32_t n_graphs, ngpde_graph_t **out);
/* The same handle built ON THE DEVICE from a COO list already in HBM (the reference moves the graph with `g |> gpu`
 * and swaps it every minibatch, docs/src/tutorials/VMH.md:132-134 + src/utils.jl:24-31): s, t are DEVICE arrays of
 * int32 or int64 (index_bits) with the given index_base.  `order` (device, n_nodes int32, nullable) is a node
 * permutation to use as the locality schedule -- e.g. the concatenated cached orders of a batch's member graphs,
 * see ngpde_graph_node_order; when NULL the lists are downloaded once and the order is computed on the host.
 * Every derived array is bit-identical to ngpde_graph_create's for the same edge list and order. Synchronises `stream`. */
int32_t ngpde_graph_create_device(int64_t n_nodes, int64_t n_edges, const void *s, const void *t, int32_t index_bits,
                                  int32_t index_base, int32_t n_graphs, const int32_t *order, ngpde_stream_t stream,
                                  ngpde_graph_t **out);
/* the handle's locality order (host buffer of n_nodes int32): cache it per graph, offset + concatenate it for batches */
int32_t ngpde_graph_node_order(const ngpde_graph_t *g, int32_t *order_out);
/* ngpde_graph_set_gcn_norm with DEVICE edge weights, built on the device (any handle) */
int32_t ngpde_graph_set_gcn_norm_device(ngpde_graph_t *g, int32_t add_self_loops, const float *edge_weight,
                                        int32_t weighted_degree, ngpde_stream_t stream);
/* Introspection of the derived arrays (device pointers; tests compare the host and device builders with it).
 * direction 0 = lists by target, 1 = by source. For NGPDE_GRAPH_HALO_OK *bytes is 1/0 and *ptr NULL. */
enum {
  NGPDE_GRAPH_ROWPTR = 0, NGPDE_GRAPH_COL = 1, NGPDE_GRAPH_EID = 2, NGPDE_GRAPH_XPOS = 3, NGPDE_GRAPH_ENT = 4,
  NGPDE_GRAPH_SCHED = 5, NGPDE_GRAPH_ELL = 6, NGPDE_GRAPH_HALO = 7, NGPDE_GRAPH_TILE_INFO = 8, NGPDE_GRAPH_SLOTS = 9,
  NGPDE_GRAPH_SLOT_W = 10, NGPDE_GRAPH_C = 11, NGPDE_GRAPH_ORDER = 12, NGPDE_GRAPH_HALO_OK = 13
};
int32_t ngpde_graph_array(const ngpde_graph_t *g, int32_t direction, int32_t which, const void **ptr, size_t *bytes);
int32_t ngpde_graph_destroy(ngpde_graph_t *g);
int32_t ngpde_graph_info(const ngpde_graph_t *g, int64_t *n_nodes, int64_t *n_edges, int32_t *n_graphs);
/* ------------------------------------------------------------------------------------------------
 * Neighbour search on the device: the COO lists that GNNGraphs.radius_graph(points, r; graph_indicator, self_loops, dir)
 * and knn_graph(points, k; ...) build on the host with NearestNeighbors.jl trees ([UPSTREAM] GraphNeuralNetworks.jl,
 * re-exported at src/NeuralGraphPDE.jl:4); the BASELINE workloads C2 and C5 are such graphs.
 *   points     device float, (dim x n) column-major = [n][dim], dim = 1, 2 or 3
 *   graph_id   device int32[n] or NULL: only points of the same graph are connected (ids id_base .. id_base+n_graphs-1)
 *   dir_out    0 = dir :in (neighbours are the sources, the point is the target; the default), 1 = :out
 *   s, t       device int32 outputs with index_base added, ordered by point; the neighbours of a point ascending by index
 *              (radius graph) or by (distance, index) (k-NN).  The reference's edge order is the tree's traversal order;
 *              the edge SET is the same.
 * Distances are sums of float squares in coordinate order without fused multiply-add; a pair is connected iff
 * d2 <= r*r (float).  ngpde_radius_graph with s == t == NULL only counts; otherwise `capacity` entries are available and
 * *n_edges (host) receives the number written.  k-NN writes exactly n*k entries; coincident points are ordered by index
 * (the reference keeps a point's own entry only while it is among the k+1 nearest).  Both synchronise `stream`.
 * ngpde_spatial_order: a node permutation (device int32[n]) along a space-filling curve through the points -- Hilbert in
 * 2-D, Morton in 3-D, the coordinate in 1-D, graph by graph -- usable as `order` of ngpde_graph_create_device, so that a
 * structure never seen before gets its locality schedule without a host traversal.
 * ---------------------------------------------------------------------------------------------- */
#define NGPDE_KNN_MAX_K 128
int32_t ngpde_radius_graph(int64_t n, int32_t dim, const float *points, float r, const int32_t *graph_id, int32_t n_graphs,
                           int32_t id_base, int32_t self_loops, int32_t dir_out, int32_t index_base, int64_t capacity, int32_t *s,
                           int32_t *t, int64_t *n_edges, ngpde_stream_t stream);
int32_t ngpde_knn_graph(int64_t n, int32_t dim, const float *points, int32_t k, const int32_t *graph_id, int32_t n_graphs,
                        int32_t id_base, int32_t self_loops, int32_t dir_out, int32_t index_base, int32_t *s, int32_t *t,
                        ngpde_stream_t stream);
int32_t ngpde_spatial_order(int64_t n, int32_t dim, const float *points, const int32_t *graph_id, int32_t n_graphs, int32_t id_base,
                            int32_t *order, ngpde_stream_t stream);

/* Device arrays of the derived graph (for callers that batch / inspect): CSR by target.
 * rowptr: int32[n_nodes+1]; col: int32[n_edges] source node of each entry; eid: int32[n_edges]
 * position of the entry in the caller's COO list. */
int32_t ngpde_graph_csr_by_target(const ngpde_graph_t *g, const int32_t **rowptr, const int32_t **col,
                                  const int32_t **eid);
int32_t ngpde_graph_csr_by_source(const ngpde_graph_t *g, const int32_t **rowptr, const int32_t **col,
                                  const int32_t **eid);

/* GCN normalisation of src/layers.jl:210-226: c = 1/sqrt(in-degree (+1 with self loops)).
 *   edge_weight       host float[n_edges] or NULL: the per-edge factor of e_mul_xj / w_mul_xj (:228,:230)
 *   weighted_degree   non-zero: degree = sum of incoming weights (explicit `edge_weight` argument,
 *                     :224 with edge_weight != nothing); zero: unweighted count (the
 *                     `use_edge_weight=true` path of the reference normalises with unweighted degrees)
 */
int32_t ngpde_graph_set_gcn_norm(ngpde_graph_t *g, int32_t add_self_loops, const float *edge_weight,
                                 int32_t weighted_degree);

/* ------------------------------------------------------------------------------------------------
 * GCNConv  -- replaces (l::GCNConv)(x, ps, st[, edge_weight]), src/layers.jl:200-239.
 *   y = act.( W * (x C (A+I) C) .+ b )     (W applied before the aggregation iff dout < din, :220)
 *   x [N][din], weight (dout x din) column-major, bias [dout] or NULL (bias=false), y [N][dout]
 *   save_agg  [N][din] or NULL: the aggregated input X C (A+I) C kept for backward (only written when dout >= din)
 *   save_z    [N][dout] or NULL: pre-activation, kept for backward
 * ---------------------------------------------------------------------------------------------- */
size_t ngpde_gcn_workspace_bytes(const ngpde_graph_t *g, int32_t din, int32_t dout, int32_t backward);
int32_t ngpde_gcn_forward(const ngpde_graph_t *g, int32_t din, int32_t dout, int32_t act,
                          const float *x, const float *weight, const float *bias, float *y,
                          float *save_agg, float *save_z, void *workspace, size_t workspace_bytes,
                          ngpde_stream_t stream);
/* Pullback of the above (what a ChainRulesCore.rrule for the shim returns; the reference gets it
 * from Zygote through gather / scatter / mul).  z: pre-activation saved by forward (for relu/identity the
 * output y may be passed instead).  saved_agg: as written by forward (ignored when dout < din).
 * dx [N][din] or NULL; dweight (dout x din) column-major; dbias [dout] or NULL.  Results overwrite. */
int32_t ngpde_gcn_backward(const ngpde_graph_t *g, int32_t din, int32_t dout, int32_t act,
                           const float *x, const float *weight, const float *z, const float *saved_agg,
                           const float *dy, float *dx, float *dweight, float *dbias, void *workspace,
                           size_t workspace_bytes, ngpde_stream_t stream);
/* ngpde_gcn_backward plus the gradient w.r.t. the `edge_weight` ARGUMENT of the call, which the reference differentiates through
 * e_mul_xj and through the weighted degree (src/layers.jl:206-231): dedge_weight[n_edges] in the caller's COO order (the ones
 * appended for the self loops are constants).  The handle must carry that call's normalisation (ngpde_graph_set_gcn_norm with
 * the weights and weighted_degree = 1).  x is always needed; bias only when dout < din (may be NULL for a layer without bias);
 * dx may be NULL.  Workspace: ngpde_gcn_backward_ew_workspace_bytes. */
size_t ngpde_gcn_backward_ew_workspace_bytes(const ngpde_graph_t *g, int32_t din, int32_t dout);
int32_t ngpde_gcn_backward_ew(const ngpde_graph_t *g, int32_t din, int32_t dout, int32_t act, const float *x, const float *weight,
                              const float *bias, const float *z, const float *saved_agg, const float *dy, float *dx, float *dweight,
                              float *dbias, float *dedge_weight, void *workspace, size_t workspace_bytes, ngpde_stream_t stream);

/* Generic aggregation  out[:, i] = sum_{e: t_e = i} w_e * x[:, s_e]  (propagate(copy_xj / w_mul_xj, g, +),
 * src/layers.jl:228-232) and its transpose (by_source != 0), any feature width d.
 * edge_weight: device float[n_edges] in COO order or NULL. aggr: NGPDE_AGGR_SUM or NGPDE_AGGR_MEAN. */
int32_t ngpde_propagate_copy_xj(const ngpde_graph_t *g, int32_t d, int32_t aggr, int32_t by_source,
                                const float *x, const float *edge_weight, float *out, ngpde_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Message-passing primitives -- the pieces `propagate(message, g, aggr; xi, xj, e)` [GraphNeuralNetworks.jl]
 * is made of, as used by ExplicitEdgeConv (src/layers.jl:94-112), VMHConv (:308-332), MPPDEConv (:390-422),
 * GNOConv (:509-547), SpectralConv (:652-657) and the GAT-style aggregation (softmax_edge_neighbors,
 * src/NeuralGraphPDE.jl:7).  Per-edge arrays are in "p order": CSR by target (ngpde_graph_csr_by_target),
 * the incoming edges of one node contiguous; ngpde_edge_permute converts from / to the COO order of
 * g.edata.  Each *_backward is the pullback Zygote would derive for the corresponding forward.
 * ---------------------------------------------------------------------------------------------- */

/* Lux Dense on a virtual vcat of up to 4 blocks:  y = act.(W * vcat(X1, X2, ...) .+ b)  (src/layers.jl:106,
 * :316, :328, :409, :418, :523 build the vcat explicitly).  Block i is [n / row_div[i]][width[i]] row-major;
 * row_div > 1 = a per-graph feature repeated over that graph's rows (repeat(theta; inner=...), :410,:418).
 * weight: (dout x sum width) column-major; save_z nullable. */
int32_t ngpde_dense_forward(int64_t n, int32_t n_seg, const float *const *seg_ptr, const int32_t *seg_width,
                            const int32_t *seg_row_div, int32_t dout, int32_t act, const float *weight,
                            const float *bias, float *y, float *save_z, ngpde_stream_t stream);
/* Up to four INDEPENDENT Dense layers (ngpde_dense_forward's arguments as tables; the segment tables of the problems follow
 * each other in seg_ptr / seg_width / seg_row_div) in ONE launch: GNOConv's node-level terms P, Q, B2 h, W h
 * (src/layers.jl:523,536) are latency-bound launches of a few dozen workgroups each. */
int32_t ngpde_dense_multi_forward(int32_t count, const int64_t *n, const int32_t *n_seg, const float *const *seg_ptr,
                                  const int32_t *seg_width, const int32_t *seg_row_div, const int32_t *dout, const int32_t *act,
                                  const float *const *weight, const float *const *bias, float *const *y, float *const *save_z,
                                  ngpde_stream_t stream);
/* Two Dense layers that share their leading input block, y_a = Dense_a(vcat(X, ...)), y_b = Dense_b(vcat(X, ...)), in one pass
 * over X: the two node-level halves of a message MLP's first layer (the `vcat(xi..., xj..., ...)` of src/layers.jl:106, :316,
 * :409-410 split into a target term and a source term).  Exactly two ngpde_dense_forward calls in effect; one launch when X
 * is 64 wide, both outputs <= 64 wide and the other blocks narrow. */
int32_t ngpde_dense_pair_forward(int64_t n, int32_t n_seg_a, const float *const *seg_ptr_a, const int32_t *seg_width_a,
                                 const int32_t *seg_row_div_a, int32_t dout_a, int32_t act_a, const float *weight_a, const float *bias_a,
                                 float *y_a, float *save_z_a, int32_t n_seg_b, const float *const *seg_ptr_b,
                                 const int32_t *seg_width_b, const int32_t *seg_row_div_b, int32_t dout_b, int32_t act_b,
                                 const float *weight_b, const float *bias_b, float *y_b, float *save_z_b, ngpde_stream_t stream);
/* Pullback of such a pair when neither layer has an activation of its own (the first-layer halves of a message MLP: the
 * activation is applied per edge) and dout = 64: dx = dy_a Wa^T + dy_b Wb^T [+ dx_addend] for the shared leading block (one
 * array, already summed; dx_addend nullable, may alias dx), both weight / bias gradients; other blocks get no gradient.
 * ngpde_dense_pair_backward_workspace_bytes returns 0 for shapes that need two ngpde_dense_backward calls instead. */
size_t ngpde_dense_pair_backward_workspace_bytes(int64_t n, int32_t n_seg_a, const float *const *seg_ptr_a, const int32_t *seg_width_a,
                                                 const int32_t *seg_row_div_a, int32_t n_seg_b, const float *const *seg_ptr_b,
                                                 const int32_t *seg_width_b, const int32_t *seg_row_div_b, int32_t dout);
int32_t ngpde_dense_pair_backward(int64_t n, int32_t n_seg_a, const float *const *seg_ptr_a, const int32_t *seg_width_a,
                                  const int32_t *seg_row_div_a, const float *weight_a, const float *dy_a, float *dweight_a,
                                  float *dbias_a, int32_t n_seg_b, const float *const *seg_ptr_b, const int32_t *seg_width_b,
                                  const int32_t *seg_row_div_b, const float *weight_b, const float *dy_b, float *dweight_b,
                                  float *dbias_b, int32_t dout, float *dx, const float *dx_addend, void *workspace,
                                  size_t workspace_bytes, ngpde_stream_t stream);
/* Chain(Dense(. => dmid, act1), Dense(dmid => dout, act2)) on a virtual vcat -- the node update psi / gamma of MPPDEConv /
 * VMHConv (src/layers.jl:418, :328).  a1 (the first layer's activations, [n][dmid]) and the save_* arrays are nullable when
 * ngpde_dense_chain2_fused(...) == 1: the intermediate then stays on chip (one launch); otherwise a1 is required. */
int32_t ngpde_dense_chain2_fused(int64_t n, int32_t n_seg, const float *const *seg_ptr, const int32_t *seg_width,
                                 const int32_t *seg_row_div, int32_t dmid, int32_t dout);
int32_t ngpde_dense_chain2_forward(int64_t n, int32_t n_seg, const float *const *seg_ptr, const int32_t *seg_width,
                                   const int32_t *seg_row_div, int32_t dmid, int32_t act1, const float *weight1, const float *bias1,
                                   float *a1, float *save_z1, int32_t dout, int32_t act2, const float *weight2, const float *bias2,
                                   float *y, float *save_z2, ngpde_stream_t stream);
size_t ngpde_dense_workspace_bytes(int64_t n, int32_t din_total, int32_t dout);
/* dseg_ptr[i] nullable: gradient block for X_i (never written for row_div > 1 blocks: graph-level features
 * are @ignore_derivatives in the reference, :397,:418).  dweight (dout x sum width) column-major, dbias nullable. */
int32_t ngpde_dense_backward(int64_t n, int32_t n_seg, const float *const *seg_ptr, const int32_t *seg_width,
                             const int32_t *seg_row_div, int32_t dout, int32_t act, const float *weight, const float *z,
                             const float *dy, float *const *dseg_ptr, float *dweight, float *dbias, void *workspace,
                             size_t workspace_bytes, ngpde_stream_t stream);

/* dst[p] = src[eid[p]] (COO order -> p order; inverse != 0: the opposite scatter), rows of d floats */
int32_t ngpde_edge_permute(const ngpde_graph_t *g, int32_t d, int32_t inverse, const float *src, float *dst,
                           ngpde_stream_t stream);

/* apply_edges for a first layer that is linear in its blocks:  z_p = P[t_p] + Q[s_p] + E_p,  a_p = act(z_p).
 * P, Q: [N][h] node-level terms (gathered at the target / source: xi = gather(., t), xj = gather(., s));
 * e_term: [E][h] in p order or NULL; outputs a_out, z_out (nullable) [E][h] in p order. */
int32_t ngpde_edge_combine_forward(const ngpde_graph_t *g, int32_t h, int32_t act, const float *p_target,
                                   const float *q_source, const float *e_term, float *a_out, float *z_out,
                                   ngpde_stream_t stream);
/* dz_p = da_p * act'(z_p) (also the gradient of e_term); dP[i] = sum over incoming edges; dQ[j] = sum over
 * outgoing edges (pullback of gather = scatter(+)).  z NULL = identity activation.  dp/dq nullable. */
int32_t ngpde_edge_combine_backward(const ngpde_graph_t *g, int32_t h, int32_t act, const float *da, const float *z,
                                    float *dz, float *dp_target, float *dq_source, ngpde_stream_t stream);

/* aggregate_neighbors(g, aggr, m): out[:, i] = aggr over the incoming edges of i of m[:, e]; mean of an empty
 * neighbourhood is 0; max/min pullback routes to every extremal entry (NNlib). */
int32_t ngpde_segment_reduce_forward(const ngpde_graph_t *g, int32_t d, int32_t aggr, const float *m, float *out,
                                     ngpde_stream_t stream);
int32_t ngpde_segment_reduce_backward(const ngpde_graph_t *g, int32_t d, int32_t aggr, const float *m, const float *out,
                                      const float *dout, float *dm, ngpde_stream_t stream);

/* GNOConv message (src/layers.jl:527-530): K_e = reshape(phi_out[:, e], cout, cin) column-major,
 * m_e = K_e * h[:, s_e].  k: [E][cin*cout] p order (element o + cout*i), h: [N][cin], m: [E][cout]. */
int32_t ngpde_gno_contract_forward(const ngpde_graph_t *g, int32_t cin, int32_t cout, const float *k, const float *h,
                                   float *m, ngpde_stream_t stream);
/* dk [E][cin*cout] nullable, dh [N][cin] nullable (needs workspace of n_edges*cin*4 bytes) */
int32_t ngpde_gno_contract_backward(const ngpde_graph_t *g, int32_t cin, int32_t cout, const float *k, const float *h,
                                    const float *dm, float *dk, float *dh, void *workspace, size_t workspace_bytes,
                                    ngpde_stream_t stream);

/* GNOConv message in reassociated form: when the last layer of phi is Dense(k => in*out) with identity activation,
 *   m_e = reshape(W2 z_e + b2, out, in) h_j = T_j z_e + Bh_j,   T_j[o][kk] = sum_i W2[o + out*i][kk] h_j[i]  (node level),
 * so the in*out x E kernel tensor of src/layers.jl:527 is never formed.  t: [N][cout*kdim] (element o*kdim + kk),
 * bh: [N][cout] or NULL, z: [E][kdim] last hidden activation of phi (p order), m: [E][cout] (p order).
 * Backward: dt [N][cout*kdim], dbh [N][cout], dz [E][kdim], each nullable. */
int32_t ngpde_gno_apply_supported(int32_t cout, int32_t kdim);
int32_t ngpde_gno_apply_forward(const ngpde_graph_t *g, int32_t cout, int32_t kdim, const float *t, const float *bh,
                                const float *z, float *m, ngpde_stream_t stream);
int32_t ngpde_gno_apply_backward(const ngpde_graph_t *g, int32_t cout, int32_t kdim, const float *t, const float *z,
                                 const float *dm, float *dt, float *dbh, float *dz, ngpde_stream_t stream);

/* The reassociated message with its per-edge input formed in the same launch:  z_e = act1(P[t_e] + Q[s_e] + E_e)  (the first Dense
 * of phi split into node-level terms, as ngpde_edge_combine_forward),  m_e = T_{s_e} z_e + Bh_{s_e}; the message is a per-source
 * GEMM on the matrix pipe (out a multiple of 16, k in {16, 32, 64}: ngpde_gno_message_supported).  z_out [E][k] (p order,
 * nullable) keeps the ACTIVATED input for the pullback, which is ngpde_gno_apply_backward followed by
 * ngpde_edge_combine_backward (for identity / relu the activated value serves as `z` there). */
int32_t ngpde_gno_message_supported(int32_t cout, int32_t kdim);
int32_t ngpde_gno_message_forward(const ngpde_graph_t *g, int32_t cout, int32_t kdim, int32_t act1, const float *p_target,
                                  const float *q_source, const float *e_term, const float *t, const float *bh, float *z_out, float *m,
                                  ngpde_stream_t stream);

/* Pullback of ngpde_gno_message_forward FOLLOWED BY the sum / mean aggregation over targets (aggregate_neighbors(g, aggr, m),
 * /root/reference/src/layers.jl:527-534), from the node-level gradient dagg [N][out]: dm_e = dagg[t_e] (sum) or dagg[t_e] /
 * deg(t_e) (mean) is formed while the per-source launch stages its rows, so the [E][out] array ngpde_segment_reduce_backward
 * would write (231 MB at BASELINE config 5, r = 0.1) is neither written nor read.  z [E][k] = z_out of the forward (the
 * ACTIVATED per-edge input), act1 identity or relu.  Outputs: dt [N][out][k], dbh [N][out] (nullable) as
 * ngpde_gno_apply_backward; dz [E][k] (p order, nullable) is the gradient of the PRE-activation, T_s^T dm_e . act1'(z_e) -- so
 * dE = dz, dP = its sums by target (ngpde_segment_reduce_forward with NGPDE_AGGR_SUM) --; dq [N][k] (nullable, needs dz) = its
 * sums by source, formed in the launch (every edge of a workgroup has the same source).  NGPDE_ERR_UNSUPPORTED for max / min /
 * mul, other activations, and shapes outside ngpde_gno_message_supported. */
int32_t ngpde_gno_message_backward_from_nodes(const ngpde_graph_t *g, int32_t cout, int32_t kdim, int32_t aggr, int32_t act1,
                                              const float *t, const float *z, const float *dagg, float *dt, float *dbh, float *dz,
                                              float *dq, ngpde_stream_t stream);

/* GNOConv's aggregated message with the EDGE index contracted first, per target over its CSR row (/root/reference/src/layers.jl:523-534;
 * the other reassociation of m_i = aggr_{e -> i} reshape(W2 z_e + b2, out, in) h_{s_e}):
 *   ngpde_gno_gform_aggregate:  z_e = act1(P[t_e] + Q[s_e] + E_e) as in ngpde_gno_message_forward;  gout [N][k][in]:
 *     G_i[k][i'] = sum_{e -> i} z_e[k] h_{s_e}[i']  (Z_i^T H_i on the matrix pipe);  hsum [N][in] (nullable) = sum_{e -> i} h_{s_e};
 *     both divided by deg(i) when `mean`;  z_out [E][k] (p order, nullable) keeps the activated input for the pullback.
 *     k = 64 and in in {32, 64, 128} (ngpde_gno_gform_supported); every array 16-byte aligned.
 *   ngpde_gno_gform_transform:  y [N][out] = act.(G W2' + hsum B2 + h W + bias)  with W2' = phi's last weight (k x in*out, Julia
 *     (in*out x k)) read as the [k*in][out] matrix it is in memory (row k*in + i', column o  <->  K_e[o, i'] = phi_out[o + out*i'],
 *     :525), the contraction split into `nsplit` slabs (ngpde_gno_gform_splits); the b2 term (hsum [N][in] with b2 read as
 *     B2[i'][o] = b2[o + out*i']; both or neither NULL) and the layer's own linear map (h [N][in], w [in][out]; both or neither NULL) are
 *     two more slabs of the SAME launch; slabs [nsplit + 2][N][out] is scratch; bias nullable; zt (nullable) keeps the pre-activation
 *     (:536).  in a multiple of 16, out of 4.
 * No [E][out] message array, no scatter, no atomics; the pullback is the by-source form's (ngpde_gno_message_backward_from_nodes). */
int32_t ngpde_gno_gform_supported(int32_t in_chs, int32_t kdim);
/* does ngpde_gno_layer_* take this form for the shape?  (inference: whenever supported; training: from about 64 edges per node, where
 * it wins back the T = W2 (x) h that the by-source pullback needs and only the by-source forward leaves behind) */
int32_t ngpde_gno_gform_preferred(int64_t n_nodes, int64_t n_edges, int32_t in_chs, int32_t kdim, int32_t cout, int32_t training);
int32_t ngpde_gno_gform_splits(int64_t n_nodes, int32_t in_chs, int32_t kdim, int32_t cout);
int32_t ngpde_gno_gform_aggregate(const ngpde_graph_t *g, int32_t in_chs, int32_t kdim, int32_t act1, int32_t mean, const float *p_target,
                                  const float *q_source, const float *e_term, const float *h, float *gout, float *hsum, float *z_out,
                                  ngpde_stream_t stream);
int32_t ngpde_gno_gform_transform(int64_t n_nodes, int32_t in_chs, int32_t kdim, int32_t cout, int32_t act, const float *gin, const float *w2,
                                  const float *hsum, const float *b2, const float *h, const float *w, const float *bias, float *y, float *zt,
                                  float *slabs, int32_t nsplit, ngpde_stream_t stream);

/* GAT-style aggregation [GraphNeuralNetworks.jl GATConv]: wx [N][heads*c] (= reshape(W x, c, heads, N)),
 * a (2c x heads) column-major; logit_e = leakyrelu(a[1:c,k].Wx[:,k,t_e] + a[c+1:2c,k].Wx[:,k,s_e]);
 * alpha = softmax over the incoming edges of each node; out[N][heads*c] = sum_e alpha_e Wx[s_e].
 * Saved for backward: alpha [E][heads] (p order), al, ar [N][heads]. */
int32_t ngpde_gat_forward(const ngpde_graph_t *g, int32_t heads, int32_t c, float negative_slope, const float *wx,
                          const float *a, float *out, float *alpha, float *al, float *ar, ngpde_stream_t stream);
size_t ngpde_gat_workspace_bytes(const ngpde_graph_t *g, int32_t heads);
int32_t ngpde_gat_backward(const ngpde_graph_t *g, int32_t heads, int32_t c, float negative_slope, const float *wx,
                           const float *a, const float *al, const float *ar, const float *alpha, const float *dout,
                           float *dwx, float *da, void *workspace, size_t workspace_bytes, ngpde_stream_t stream);

/* The whole GAT-style layer  y = act.(GAT(x) .+ b)  with W (heads*c x din), a (2c x heads), heads concatenated, in ONE launch
 * (pullback: two launches + one reduction) when din == heads * c == 64, heads in {1, 2, 4} and the graph's tiles fit the LDS halo
 * in both directions (ngpde_gat_layer_supported; BASELINE config 3 = 64 => 4 x 16): W x of the rows a tile stages (its own and its
 * halo) is formed on the matrix pipe inside the launch, logits and messages work on it per head; no W x array exists in memory.  Self loops are edges of g (the caller appends them, as GATConv's add_self_loops does).
 *   save_alpha [E][heads] nullable: attention coefficients in p order, the sign bit carrying leakyrelu's branch (pullback only)
 *   save_z     [N][64] nullable: pre-activation (the pullback of activations other than identity / relu needs it)
 * backward: y_or_z = y for relu, z otherwise (ignored for identity); dx nullable; dweight (64 x 64) column-major, da (2c x heads),
 * dbias [64] nullable.  workspace: ngpde_gat_layer_workspace_bytes. */
int32_t ngpde_gat_layer_supported(const ngpde_graph_t *g, int32_t din, int32_t heads, int32_t c);
size_t ngpde_gat_layer_workspace_bytes(const ngpde_graph_t *g, int32_t heads, int32_t c);
int32_t ngpde_gat_layer_forward(const ngpde_graph_t *g, int32_t din, int32_t heads, int32_t c, float negative_slope, int32_t act,
                                const float *x, const float *weight, const float *a, const float *bias, float *y,
                                float *save_alpha, float *save_z, ngpde_stream_t stream);
int32_t ngpde_gat_layer_backward(const ngpde_graph_t *g, int32_t din, int32_t heads, int32_t c, float negative_slope, int32_t act,
                                 const float *x, const float *weight, const float *a, const float *y_or_z, const float *save_alpha,
                                 const float *dy, float *dx, float *dweight, float *da, float *dbias, void *workspace,
                                 size_t workspace_bytes, ngpde_stream_t stream);

/* y = act.(a .+ addend .+ b): the tail of a layer whose linear part was computed elsewhere -- GNOConv's
 * sigma(W x + m + b) (src/layers.jl:536-547) with a = aggregated messages, addend = W x; the GAT-style layer's bias + activation.
 * a, y [n][d]; addend [n][d] nullable; bias [d] nullable; save_z nullable.  Backward: dz = dy .* act'(z) (the gradient of a AND
 * of addend; for the identity dz may alias dy and the pass is skipped), dbias = column sums of dz (nullable; needs the workspace). */
int32_t ngpde_bias_act_forward(int64_t n, int32_t d, int32_t act, const float *a, const float *addend, const float *bias, float *y,
                               float *save_z, ngpde_stream_t stream);
size_t ngpde_bias_act_workspace_bytes(int32_t d);
int32_t ngpde_bias_act_backward(int64_t n, int32_t d, int32_t act, const float *dy, const float *z, float *dz, float *dbias,
                                void *workspace, size_t workspace_bytes, ngpde_stream_t stream);

/* Fused message path  m_i = aggr_{e: t_e = i} phi(...)  for a message MLP whose first layer has been split into
 * node-level terms (ngpde_edge_combine_forward) and whose remaining layers are Dense, all widths <= 64 and
 * multiples of 4, at most 3 layers after the first (e.g. MPPDEConv 132 => 64 => 64, src/layers.jl:402-416; the VMH
 * tutorial's 4 => 60 => 60 => 60 => 40, docs/src/tutorials/VMH.md:75-79): gather through LDS, MFMA layers with
 * LDS-resident weights, in-tile segmented reduction; no per-edge array is written unless save_z[l] is non-NULL
 * (save_z[0]: z1 [E][h1]; save_z[l]: pre-activation of layer l [E][tail_dout[l-1]], p order) for the pullback.
 * Needs ngpde_graph_set_gcn_norm to have been called (it builds the tile schedule) and a graph whose tiles fit
 * the LDS halo; ngpde_edge_mlp_supported returns 1 when this entry can be used, otherwise compose the
 * primitives above.  aggr: + / mean / max / min / * (NGPDE_AGGR_MUL: an empty neighbourhood gives 1, like scatter(*)).
 * out: [N][last width]. */
int32_t ngpde_edge_mlp_supported(const ngpde_graph_t *g, int32_t h1, int32_t n_tail, const int32_t *tail_dout);
int32_t ngpde_edge_mlp_forward(const ngpde_graph_t *g, int32_t h1, int32_t act1, const float *p_target,
                               const float *q_source, const float *e_term, int32_t n_tail, const int32_t *tail_dout,
                               const int32_t *tail_act, const float *const *tail_weight, const float *const *tail_bias,
                               int32_t aggr, float *out, float *const *save_z, ngpde_stream_t stream);
/* Fused pullback of the same message path for 0 or 1 Dense layer after the first and any of the five aggregations (*: a first pass over
 * a tile's edges forms each target's product of the nonzero messages and the number of zeros; every message then receives the gradient
 * times the product of the OTHERS -- the one zero message of a row the product of the rest, nothing where two are zero; max / min: the
 * first pass leaves the target's extremum and every message equal to it receives the gradient, as NNlib's scatter pullback): recomputes the
 * per-edge activations inside the tile (nothing per-edge has to be saved by the forward), accumulates the tail layer's
 * weight / bias gradient on MFMA in per-workgroup slabs, writes dz1 once ([E][h1], p order: de_term, required -- it is the
 * gradient of the per-edge first-layer term and the input of the by-source sum that gives dq_source) and sums it per target
 * into dp_target.  dout: [N][last width] gradient of the aggregated messages.
 * Without a per-edge first-layer term (e_term NULL: MPPDEConv on graphs without edge features, src/layers.jl:407-410) the
 * 64-wide two-layer specialisation sums dz1 by source inside its launch where every tile's halo has at most 48 rows (meshes):
 * de_term may then be NULL and no [E][h1] array exists at all (ngpde_edge_mlp_backward_needs_edge_buffer returns 0). */
int32_t ngpde_edge_mlp_backward_supported(const ngpde_graph_t *g, int32_t h1, int32_t n_tail, const int32_t *tail_dout, int32_t aggr);
int32_t ngpde_edge_mlp_backward_needs_edge_buffer(const ngpde_graph_t *g, int32_t h1, int32_t act1, int32_t has_e_term, int32_t n_tail,
                                                   const int32_t *tail_dout, const int32_t *tail_act, int32_t aggr);
size_t ngpde_edge_mlp_backward_workspace_bytes(const ngpde_graph_t *g, int32_t h1, int32_t n_tail, const int32_t *tail_dout);
int32_t ngpde_edge_mlp_backward(const ngpde_graph_t *g, int32_t h1, int32_t act1, const float *p_target, const float *q_source,
                                const float *e_term, int32_t n_tail, const int32_t *tail_dout, const int32_t *tail_act,
                                const float *const *tail_weight, const float *const *tail_bias, int32_t aggr, const float *dout,
                                float *dp_target, float *dq_source, float *de_term, float *const *dtail_weight,
                                float *const *dtail_bias, void *workspace, size_t workspace_bytes, ngpde_stream_t stream);
/* a = act.(z) (re-materialises an activation from a saved pre-activation in the pullback of the fused path) */
int32_t ngpde_activation_forward(int64_t count, int32_t act, const float *z, float *a, ngpde_stream_t stream);

/* SpectralConv message weights (src/layers.jl:654): w_e = cos(e n / 2) cot(e / 2) / 2; the layer is then
 * propagate(e_mul_xj, g, +) = ngpde_propagate_copy_xj with these weights. */
int32_t ngpde_spectral_weights(int64_t n_edges, int32_t n, const float *e, float *w, ngpde_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fixed-step neural graph ODE over  Chain(GCNConv(d => d, act), GCNConv(d => d, act))  -- the caller
 * of the hot path in the reference's tutorial (docs/src/tutorials/graph_node.md:44-66, :78): the
 * right-hand side dudt(u, p, t) is evaluated once per Runge-Kutta stage.  BASELINE configs fix the
 * step count (Euler x 10, Tsit5 x 50), so the integrator is fixed-step and the backward pass is the
 * discrete adjoint (backprop through the steps), with the whole solve replayed from one HIP graph.
 * d in {16, 32, 64, 128}.  The plan owns its tape (saved activations) and scratch buffers.
 * ---------------------------------------------------------------------------------------------- */
int32_t ngpde_node_gcn2_create(const ngpde_graph_t *g, int32_t d, int32_t act, int32_t tableau,
                               int32_t n_steps, float dt, int32_t with_backward, ngpde_node_t **out);
/* The same plan for a block-diagonal batch of `members` graphs that all have the structure of `member` ("all graphs need to
 * have the same structure", src/layers.jl:359-361; MLUtils.batch, test/runtests.jl:89-102): u0 / uT / duT / du0 are
 * [members * N][d], member after member; the parameter gradients are the sums over the members.  The trajectories are solved
 * one after the other inside the persistent launches (a tile keeps a trajectory's state in registers, see ngpde_node_flags), so
 * the derived-graph handle, the wait lists and the exchanged arrays are those of ONE member.  NGPDE_ERR_UNSUPPORTED when the
 * persistent plan is not available for this graph / width / activation: batch the graphs into one handle then. */
int32_t ngpde_node_gcn2_create_batch(const ngpde_graph_t *member, int32_t members, int32_t d, int32_t act, int32_t tableau,
                                     int32_t n_steps, float dt, int32_t with_backward, ngpde_node_t **out);
int32_t ngpde_node_destroy(ngpde_node_t *plan);
size_t ngpde_node_tape_bytes(const ngpde_node_t *plan);
/* u0 [N][d]; w1,w2 (d x d) column-major; b1,b2 [d]; uT [N][d].  Enqueues the whole solve. */
int32_t ngpde_node_gcn2_forward(ngpde_node_t *plan, const float *u0, const float *w1, const float *b1,
                                const float *w2, const float *b2, float *uT, ngpde_stream_t stream);
/* duT: adjoint of u(T).  Outputs: du0 [N][d], dw1,dw2 (d x d) column-major, db1,db2 [d]. */
int32_t ngpde_node_gcn2_backward(ngpde_node_t *plan, const float *duT, float *du0, float *dw1,
                                 float *db1, float *dw2, float *db2, ngpde_stream_t stream);
/* ONE solve in flight per plan: the plan owns a single tape, u0 copy and parameter copies, so a second
 * ngpde_node_gcn2_forward overwrites what the first one's backward needs.  Every forward stamps a new generation (1, 2, ...);
 * a caller that may interleave solves (an autograd tape holding two forwards) records the generation after its forward and
 * calls ngpde_node_expect_generation right before its backward: NGPDE_ERR_STATE when another forward has run on the plan
 * since.  *backward_pending = 1 while a forward of a with_backward plan has not been followed by its backward (take another
 * plan for the next solve then).  A plan must be used from one stream at a time: its buffers are shared by whatever streams
 * the calls are given and carry no cross-stream ordering. */
int32_t ngpde_node_generation(const ngpde_node_t *plan, uint64_t *generation, int32_t *backward_pending);
int32_t ngpde_node_expect_generation(const ngpde_node_t *plan, uint64_t generation);
/* Names and average device time of the plan's kernels are visible to rocprofv3 --kernel-trace;
 * this returns the number of kernel launches one forward (+ backward) solve enqueues. */
int32_t ngpde_node_launch_count(const ngpde_node_t *plan, int32_t *forward, int32_t *backward);
/* which internal forms the plan chose: bit 0 = pre-scaled arrays (rows held as c .* x, halo rows staged by LDS-DMA),
 * bit 1 = relu sign-bit masks instead of saved layer outputs, bit 2 = eager launches (no HIP-graph replay), bits 3 / 4 = the
 * forward solve / the adjoint run as ONE persistent launch each, tiles synchronised inside the launch by per-tile phase flags:
 * graphs whose tiles fit the LDS halo, d = 64 (d = 16 / 32 zero-padded onto it, bit 7), any activation; up to 2 tiles per
 * co-resident workgroup in registers (bit 5), up to 8 taking turns (bit 6); graphs with edge weights on the turn-taking form with
 * the slot weights in LDS (up to 3 tiles per workgroup); graphs with hubs of at most one tile per CU in the hub geometry (bit 8).  d = 128,
 * larger graphs with hubs keep the replayed plan (edge weights and batches of same-structure graphs with hubs run in the hub geometry).  A persistent launch
 * needs all its workgroups resident at once: run one such solve at a time per device (NGPDE_NO_PERSISTENT=1 selects the
 * replayed plan otherwise).  Its waits are bounded; a launch that gives up writes NaN outputs and raises the plan's fault
 * flag, which ngpde_node_fault reads (synchronises `stream`). */
enum { NGPDE_NODE_PRESCALED = 1, NGPDE_NODE_SIGN_MASKS = 2, NGPDE_NODE_EAGER = 4, NGPDE_NODE_PERSISTENT_FWD = 8,
       NGPDE_NODE_PERSISTENT_BWD = 16, NGPDE_NODE_TILE_PAIRS = 32 /* persistent launches with two tiles per workgroup */,
       NGPDE_NODE_TILE_ROUNDS = 64 /* persistent launches with k tiles per workgroup taking turns (larger graphs) */,
       NGPDE_NODE_WIDENED = 128 /* d = 16 / 32 run zero-padded on the 64-wide persistent kernels (NGPDE_NO_WIDEN=1 turns it off) */,
       NGPDE_NODE_HUB_GEOMETRY = 256 /* persistent launches in the hub geometry: graphs of at most one 32-row tile per CU whose tiles reach
                                        beyond the 96-row halo / 32-entry rows (a Cora-shaped graph, docs/src/tutorials/graph_node.md:14-23):
                                        256-row halos, variable-length rows, hub rows summed by all lane groups of the workgroup */,
       NGPDE_NODE_OWN_FIRST = 512 /* the plan reads its own slot tables (by-target lists): each row's own-tile neighbours first, which the
                                     one-tile forward launch sums while it waits for the neighbouring tiles; every kernel of the plan
                                     sums in that order (graphs of at most two tiles per CU; NGPDE_NO_OWN_FIRST=1 at create keeps the handle's order) */ };
int32_t ngpde_node_flags(const ngpde_node_t *plan, int32_t *flags);
/* Host only, no device call: the node numbering of a block-diagonal batch (Flux.batch / MLUtils.batch of single graphs,
 * /root/reference/test/runtests.jl:89-102, docs/src/tutorials/VMH.md:120-134) whose members are padded to whole 32-row tiles, so
 * that a solver plan for `n_members` identical-shape members (ngpde_ode_create, members > 1) or the tile kernels can take it:
 * member k keeps its node order and is followed by isolated nodes up to the next multiple of 32.  sizes[k] = nodes of member k.
 * Out: padded_offsets[k] = first node of member k in the padded batch (n_members + 1 entries, the last = padded node count);
 * index[v] = position of node v of the unpadded batch (sum of sizes entries) -- the gather / scatter index of
 * ngpde_rows_index; if `order` (a locality order of the unpadded batch, each member's nodes contiguous) is given, order_padded
 * (padded node count entries) = the same order with every member's padding nodes behind it.  NGPDE_ERR_INVALID_ARGUMENT for a
 * negative size or an `order` that is no such permutation. */
int32_t ngpde_batch_pad_host(int32_t n_members, const int64_t *sizes, int64_t *padded_offsets, int64_t *index, const int32_t *order,
                             int32_t *order_padded);
/* Host only, no device call: would a graph with hubs run on the persistent solver's hub geometry (NGPDE_NODE_HUB_GEOMETRY)?  The
 * graph is given as its two 0-based CSR lists (by target: the in-neighbours of every node; by source: the out-neighbours).  On
 * NGPDE_OK order[32 t + k] is the node the solver would put in row k of tile t (n_nodes entries) and, if tile_rows is not NULL,
 * tile_rows[2 t + dir] the distinct rows tile t references through list dir (its own rows included; at most 256).
 * NGPDE_ERR_UNSUPPORTED (text in ngpde_last_error) when the graph has more than 256 tiles, a node of more than ~224 distinct in+out
 * neighbours, or a tile of more than 4 096 list entries: the solver then replays its per-stage launches.  [no reference counterpart:
 * the reference evaluates Cora (docs/src/tutorials/graph_node.md:14-23) with the generic gather / scatter] */
int32_t ngpde_hub_partition_host(int64_t n_nodes, const int32_t *rowptr_by_target, const int32_t *col_by_target,
                                 const int32_t *rowptr_by_source, const int32_t *col_by_source, int32_t *order, int32_t *tile_rows);
int32_t ngpde_node_fault(ngpde_node_t *plan, ngpde_stream_t stream, int32_t *fault);
/* Diagnostic of the interleaved batch solve (ngpde_node_gcn2_create_batch, two members per workgroup): of the `slot_phases`
 * (tile, member, phase) units of the last forward / adjoint launch, how many found their halo rows gathered ahead of time, i.e.
 * paid no exposed hand-off.  Zeros for plans without persistent launches.  Synchronises `stream`.  [no reference counterpart:
 * the batch is test/runtests.jl:89-102] */
int32_t ngpde_node_pipeline_stats(ngpde_node_t *plan, ngpde_stream_t stream, int64_t *ahead_forward, int64_t *ahead_backward,
                                  int64_t *slot_phases);
/* The same solve with ONE GAT-style layer (ngpde_gat_layer_*: din = heads * c = 64, heads in {1, 2, 4}, weight (64 x 64), a (2c x
 * heads), bias [64] nullable, activation `act`) as the right-hand side  du/dt = act.(GAT(u) .+ b): one persistent launch for the
 * forward solve, one for the discrete adjoint, per-tile phase flags between neighbouring tiles as in the GCN solver.  Runs the
 * one-launch layer's own per-tile code and combines the stages with the coefficients float(dt * a_ij) in the order of
 * ngpde_rk_stage_combine, so u(T) and du0 are bitwise those of the generic solver over ngpde_gat_layer_forward / _backward;
 * parameter gradients agree to rounding (summed per tile over the whole adjoint).  [BASELINE config 3 "GAT as ODE RHS";
 * /root/reference/docs/src/tutorials/VMH.md:85-89 NeuralODE(layer); softmax_edge_neighbors: src/NeuralGraphPDE.jl:7]
 *   create:   ERR_UNSUPPORTED when ngpde_node_gat_supported is 0 (other widths / head counts, tiles that do not fit the LDS halo,
 *             more tiles than the device keeps resident, NGPDE_NO_PERSISTENT=1): use the generic solver.
 *   forward:  u0, uT [N][64]; with_backward plans keep the tape (stage inputs, y / z, alpha: ngpde_node_gat_tape_bytes).
 *   backward: duT [N][64] -> du0 [N][64], dweight (64 x 64, layout of weight), da (2c x heads), dbias [64] nullable.
 *   fault:    1 when a launch gave up waiting (outputs NaN); the plan then refuses further launches (ERR_STATE).
 *   supported: compares the two directions' tile schedules on the device and SYNCHRONISES: ask once per graph handle, not per solve. */
int32_t ngpde_node_gat_supported(const ngpde_graph_t *g, int32_t din, int32_t heads, int32_t c);
int32_t ngpde_node_gat_create(const ngpde_graph_t *g, int32_t heads, int32_t c, float negative_slope, int32_t act, int32_t tableau,
                              int32_t n_steps, double dt, int32_t with_backward, ngpde_node_gat_t **out);
/* A block-diagonal batch of `members` identical structures (g is ONE member; test/runtests.jl:89-102): u0 / uT / duT / du0 are
 * [members][N][64], the parameter gradients the sum over the members; two members at a time share a workgroup, one computing while
 * the other's rows and flags travel. */
int32_t ngpde_node_gat_create_batch(const ngpde_graph_t *g, int32_t members, int32_t heads, int32_t c, float negative_slope,
                                    int32_t act, int32_t tableau, int32_t n_steps, double dt, int32_t with_backward,
                                    ngpde_node_gat_t **out);
int32_t ngpde_node_gat_destroy(ngpde_node_gat_t *plan);
size_t ngpde_node_gat_tape_bytes(const ngpde_node_gat_t *plan);
int32_t ngpde_node_gat_fault(ngpde_node_gat_t *plan, ngpde_stream_t stream, int32_t *fault);
int32_t ngpde_node_gat_forward(ngpde_node_gat_t *plan, const float *u0, const float *weight, const float *a, const float *bias,
                               float *uT, ngpde_stream_t stream);
int32_t ngpde_node_gat_backward(ngpde_node_gat_t *plan, const float *weight, const float *a, const float *duT, float *du0,
                                float *dweight, float *da, float *dbias, ngpde_stream_t stream);
/* ---- NeuralODE(VMHConv(phi, gamma)) device-resident (node_vmh.hip) -----------------------------------------------------------
 * The reference's second neural-ODE caller: docs/src/tutorials/VMH.md:75-89 -- du/dt = VMHConv(phi, gamma)(u) (src/layers.jl:308-332:
 * m_i = aggr_j phi([h_i; h_j - h_i; x_j - x_i]), h' = gamma([h_i; m_i])) under a fixed-step solver, forward + discrete adjoint as ONE
 * persistent launch each (+ one weight-pullback GEMM per Dense layer).  Shapes taken: a scalar state (hd = 1: u0, uT, duT, du0 are
 * [N]), pd = 1..3 position coordinates (pos: device [N][pd], copied at creation), MLPs of 2..4 Dense layers up to 64 wide --
 * phi: dims[0] = 2 hd + pd, gamma: dims[0] = hd + phi's output width, gamma's output width = hd --, hidden activations identity /
 * relu / tanh / sigmoid, identity output layers, + or mean aggregation, graphs of up to 64 tiles per resident workgroup (more half tiles than resident
 * workgroups: whole 32-row tiles, K per workgroup, walked in turn in every phase -- "tile rounds").
 * ngpde_node_vmh_supported says so; the host's generic solver takes everything else.  Weights are [in][out] (Julia's (out x in)
 * column-major), passed as HOST arrays of device pointers (bias pointers / the bias arrays may be NULL).  One solve's tape per plan. */
int32_t ngpde_node_vmh_supported(const ngpde_graph_t *g, int32_t hd, int32_t pd, int32_t n_phi, const int32_t *phi_dims, const int32_t *phi_acts,
                                 int32_t n_gamma, const int32_t *gamma_dims, const int32_t *gamma_acts, int32_t aggr);
int32_t ngpde_node_vmh_create(const ngpde_graph_t *g, int32_t hd, int32_t pd, const float *pos, int32_t n_phi, const int32_t *phi_dims,
                              const int32_t *phi_acts, int32_t n_gamma, const int32_t *gamma_dims, const int32_t *gamma_acts, int32_t aggr,
                              int32_t tableau, int32_t n_steps, double dt, int32_t with_backward, ngpde_node_vmh_t **out);
int32_t ngpde_node_vmh_destroy(ngpde_node_vmh_t *plan);
size_t ngpde_node_vmh_tape_bytes(const ngpde_node_vmh_t *plan);
/* The tapes of destroyed VMH plans are parked for the next plan they fit (a training loop that re-batches its point clouds every epoch,
 * VMH.md:120-141, builds a plan per step; blocks of tens of GB cost seconds to allocate and free).  Gives them back to the device;
 * returns the bytes released.  The library does it by itself before it reports that a plan's tapes do not fit.  [no reference
 * counterpart: CUDA.reclaim() is the caller-side analogue] */
size_t ngpde_release_cached_memory(void);
int32_t ngpde_node_vmh_fault(ngpde_node_vmh_t *plan, ngpde_stream_t stream, int32_t *fault);
int32_t ngpde_node_vmh_forward(ngpde_node_vmh_t *plan, const float *u0, const float *const *phi_weight, const float *const *phi_bias,
                               const float *const *gamma_weight, const float *const *gamma_bias, float *uT, ngpde_stream_t stream);
int32_t ngpde_node_vmh_backward(ngpde_node_vmh_t *plan, const float *const *phi_weight, const float *const *gamma_weight, const float *duT,
                                float *du0, float *const *dphi_weight, float *const *dphi_bias, float *const *dgamma_weight,
                                float *const *dgamma_bias, ngpde_stream_t stream);
/* saveat (docs/src/tutorials/VMH.md:85 `NeuralODE(gnn, tspan, Tsit5(); saveat = dt_train)`, the loss of :104-108 reads every saved
 * state): usave / dusave are [T][N], T = n_steps / save_every + (save_start ? 1 : 0), slot 0 = u0 when save_start, the last slot =
 * u(T); save_every must divide the plan's steps.  The adjoint adds dusave[j] to lambda at the time of state j; du0 must not lie
 * inside dusave.  (forward / backward above are the T = 1 case.) */
int32_t ngpde_node_vmh_forward_saveat(ngpde_node_vmh_t *plan, const float *u0, const float *const *phi_weight, const float *const *phi_bias,
                                      const float *const *gamma_weight, const float *const *gamma_bias, int32_t save_every,
                                      int32_t save_start, float *usave, ngpde_stream_t stream);
int32_t ngpde_node_vmh_backward_saveat(ngpde_node_vmh_t *plan, const float *const *phi_weight, const float *const *gamma_weight,
                                       int32_t save_every, int32_t save_start, const float *dusave, float *du0, float *const *dphi_weight,
                                       float *const *dphi_bias, float *const *dgamma_weight, float *const *dgamma_bias,
                                       ngpde_stream_t stream);

/* Measurement aid (not on the product path): re-runs the last solve (forward, and backward of
 * loss = sum(u(T)) when the plan has one) launch by launch with start/stop events attached to every
 * `stride`-th dispatch and returns the mean DEVICE time per launch in microseconds -- the quantity
 * rocprofv3 --kernel-trace reports -- for the four kernel roles:
 *   out_us[0] forward layer 1, out_us[1] forward layer 2 + stage combination,
 *   out_us[2] backward layer 1, out_us[3] backward stage combination + layer 2.
 * A persistent plan has one launch per direction: out_us[0] = the forward solve, out_us[2] = the adjoint, the others 0.
 * out_count[4] (nullable) receives the number of sampled launches per role.  Synchronises `stream`. */
int32_t ngpde_node_profile(ngpde_node_t *plan, int32_t stride, float *out_us, int32_t *out_count,
                           ngpde_stream_t stream);

/* Runge-Kutta combination for right-hand sides evaluated by ARBITRARY layers (a GAT-style layer, NeuralODE(VMHConv) of
 * docs/src/tutorials/VMH.md:85-89, ...):  out = c_self * base + sum_k coefs[k] * terms[k]  over `count` floats, n_terms <= 8
 * (Tsit5 has at most six stage terms).  It is the stage input u + dt sum_j a_ij k_j, the step update u + dt sum_i b_i k_i, and in
 * the discrete adjoint K-bar_i = dt b_i lambda + dt sum_{j>i} a_ji U-bar_j, lambda += sum_j U-bar_j and the accumulation of the
 * parameter gradients over the stages.  terms / coefs are HOST arrays (of device pointers / floats); base may be NULL
 * (c_self ignored); out may alias base or a term.  One launch, no allocation: graph-capture safe. */
int32_t ngpde_rk_stage_combine(int64_t count, float c_self, const float *base, int32_t n_terms, const float *const *terms,
                               const float *coefs, float *out, ngpde_stream_t stream);

/* acc[k][i] += g[k][i], i < counts[k], for n_arrays arrays in one launch (per 24 arrays): the cotangents of ALL parameters of one
 * right-hand-side pullback added to their accumulators over the stages of the discrete adjoint (the same arithmetic as
 * ngpde_rk_stage_combine(count, 1, acc, 1, {g}, {1}, acc) per array: fma(1, g, 1 * acc)).  acc / g / counts are HOST arrays. */
int32_t ngpde_accumulate_many(int32_t n_arrays, float *const *acc, const float *const *g, const int64_t *counts, ngpde_stream_t stream);

/* Optimiser step on the flat parameter vector, one launch behind the gradient all-reduce on the same stream
 * [UPSTREAM Optimisers.jl Adam / Rprop; reference call sites docs/src/tutorials/graph_node.md:90,122-129, VMH.md:97].
 * grad_scale multiplies the (reduced) gradient first: 1/world_size for a mean over data-parallel ranks.
 * Adam: step counts from 1.  Rprop: grad_prev starts at 0, step_size at eta. */
int32_t ngpde_adam_step(int64_t n, float *x, const float *grad, float *m, float *v, float eta, float beta1, float beta2,
                        float eps, int64_t step, float grad_scale, ngpde_stream_t stream);
int32_t ngpde_rprop_step(int64_t n, float *x, const float *grad, float *grad_prev, float *step_size, float shrink, float grow,
                         float step_min, float step_max, float grad_scale, ngpde_stream_t stream);

/* ---- data-parallel collective (comm.hip) -----------------------------------------------------------------------------------
 * New functionality (the reference has no multi-GPU code): whole trajectories / graphs of a batch shard across one process per
 * GPU (test/runtests.jl:89-102, src/layers.jl:359-361), parameters are replicated, and the flat parameter-gradient vector is
 * summed over the ranks once per backward pass -- ncclAllReduce(sum, fp32) over RCCL / xGMI on the caller's stream, in place --
 * optionally with the fused Adam step (1 / world folded in) as the next launch on that stream (graph_node.md:122-129).
 *   ngpde_comm_unique_id: rank 0 makes the id (NGPDE_COMM_ID_BYTES bytes); the HOST ships it to the other ranks.
 *   ngpde_comm_create:    every rank, on its current device; collective (returns when all ranks have joined).
 * RCCL is looked up at the first call: NGPDE_ERR_UNSUPPORTED when librccl.so is absent.  One rank per device (RCCL refuses two). */
typedef struct ngpde_comm ngpde_comm_t;
#define NGPDE_COMM_ID_BYTES 128
int32_t ngpde_comm_unique_id(void *id_out, size_t id_bytes);
int32_t ngpde_comm_create(const void *unique_id, int32_t rank, int32_t world, ngpde_comm_t **out);
int32_t ngpde_comm_destroy(ngpde_comm_t *comm);
int32_t ngpde_comm_info(const ngpde_comm_t *comm, int32_t *rank, int32_t *world);
/* what RCCL reports about the communicator (ncclCommCount, ncclCommUserRank): evidence in a run's record that `count` ranks met */
int32_t ngpde_comm_rccl_info(const ngpde_comm_t *comm, int32_t *count, int32_t *user_rank);
int32_t ngpde_grad_allreduce(ngpde_comm_t *comm, float *flat, int64_t count, ngpde_stream_t stream);
int32_t ngpde_grad_allreduce_adam(ngpde_comm_t *comm, int64_t n, float *x, float *grad, float *m, float *v, float eta, float beta1,
                                  float beta2, float eps, int64_t step, ngpde_stream_t stream);

/* ---- weight-sized rearrangements around the edge-function layers (row_blocks.hip) ------------------------------------------
 * The first Dense layer of phi is split by ROW BLOCKS of its [in][out] weight and the blocks are recombined with signs
 * (src/layers.jl:106 ExplicitEdgeConv, :316 VMHConv, :409-410 MPPDEConv, :523 GNOConv).  ngpde_row_blocks_gather builds up to
 * four recombined matrices in ONE launch: output o, rows dst_row0[s] .. + n_rows[s], += sign[s] * src rows src_row0[s] .. (rows no
 * segment covers are zero; overlapping segments add, e.g. VMHConv's wa - wb).  ngpde_row_blocks_scatter is its pullback: the
 * gradient of the source from the gradients of the outputs (a NULL entry of `douts` counts as zero), every source row written. */
int32_t ngpde_row_blocks_gather(int32_t width, int32_t src_rows, const float *src, int32_t n_seg, const int32_t *out_index,
                                const int32_t *dst_row0, const int32_t *src_row0, const int32_t *n_rows, const float *sign,
                                int32_t n_out, float *const *outs, const int32_t *out_rows, ngpde_stream_t stream);
int32_t ngpde_row_blocks_scatter(int32_t width, int32_t src_rows, float *dsrc, int32_t n_seg, const int32_t *out_index,
                                 const int32_t *dst_row0, const int32_t *src_row0, const int32_t *n_rows, const float *sign,
                                 int32_t n_out, float *const *douts, const int32_t *out_rows, ngpde_stream_t stream);
/* dst [cols][rows] = transpose of src [rows][cols] (GNOConv's reassociated form reads phi's last weight transposed,
 * src/layers.jl:527-530; its pullback transposes the gradient back) */
int32_t ngpde_transpose(int32_t rows, int32_t cols, const float *src, float *dst, ngpde_stream_t stream);
/* Rows by an index list (int64, device, 0-based; entries distinct and in [0, n_rows) -- an entry outside that range names no row: the
 * gather writes a zero row for it, the scatter skips it): gather  dst[o][i][:] = src[o][index[i]][:]  (src [outer][n_rows][d],
 * dst [outer][n_index][d]) or, scatter != 0, dst[o][index[i]][:] = src[o][i][:] with every other row of dst zero (each is the other's
 * pullback).  The state of a batch of point clouds whose members were padded to whole tiles goes in and out of the device-resident
 * NeuralODE(VMHConv) plan through it (docs/src/tutorials/VMH.md:120-134: the batch is one block-diagonal graph). */
int32_t ngpde_rows_index(int64_t outer, int64_t n_rows, int64_t n_index, int32_t d, const int64_t *index, const float *src, float *dst,
                         int32_t scatter, ngpde_stream_t stream);
/* out[i][:] = x[i][:] * scale[i]  (mean aggregation's 1 / degree applied once per node to a cotangent, :534) */
int32_t ngpde_rows_scale(int64_t n, int32_t d, const float *x, const float *scale, float *out, ngpde_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Layer-level entries: ONE call evaluates an edge-function layer of the reference, one call its pullback.
 *   ExplicitEdgeConv  h'_i = aggr_j phi([h_i; h_j; x_j - x_i])                                    /root/reference/src/layers.jl:94-112
 *   VMHConv           m_i = aggr_j phi([h_i; h_j - h_i; x_j - x_i]),  h'_i = gamma([h_i; m_i])     :308-332
 *   MPPDEConv         m_i = aggr_j phi([h_i; h_j; d_i - d_j; e_ij; theta]),  h'_i = psi([h_i; m_i; theta])   :390-422
 * The caller describes the layer -- the blocks the reference vcats and the Dense stacks phi / update (gamma, psi) -- and the
 * library does what the host side had to orchestrate before: the split of phi's first weight into a target-side and a
 * source-side matrix (the signed row blocks of :106, :316, :409-410), the node-level terms P, Q (one pass over h when the shape
 * allows) and the per-edge term of the edge features, the message path (one fused launch where the message MLP fits the fused
 * kernels, the gather / Dense / segmented-reduce primitives otherwise), the node update as a chained launch, and in the pullback
 * the same steps in reverse with every temporary and saved activation laid out in ONE caller-provided workspace.
 *
 * Blocks are row-major device arrays (= the reference's column-major matrices):
 *   state[k] [N][state_width[k]]   the values of `x` (an array = one block; a NamedTuple = its values in order, :94-96, :308-310);
 *                                  MPPDEConv takes exactly one (h)
 *   node_feat [N][node_feat_width] EdgeConv / VMH: vcat of g.ndata WITHOUT :x (constants, concatenated behind the state on both
 *                                  sides of the message, :104, :315, :324); MPPDE: vcat(values(g.ndata)...) = d (:403-405)
 *   pos [N][pos_width]             EdgeConv / VMH: g.ndata.x (x_j - x_i);  MPPDE: unused
 *   edge_feat [E][edge_feat_width] MPPDE: vcat(values(g.edata)...) in p order (ngpde_edge_permute), nullable (:407)
 *   theta [G][theta_width]         MPPDE: vcat(values(g.gdata)...) per graph of the batch, repeated over each graph's nodes / edges
 *                                  (:397, :410, :418); the batch's graphs share one structure (:359-361)
 * Only `state` carries a gradient (graph features are closed-over constants, theta is @ignore_derivatives).
 * A Dense stack: n_layers in 1..NGPDE_MLP_MAX_LAYERS, layer l maps dims[l] => dims[l + 1] with activation act[l], weight[l]
 * (dims[l + 1] x dims[l]) column-major, bias[l] nullable.  update.n_layers = 0 for ExplicitEdgeConv.
 *
 * forward:  y [N][out]; `training` != 0 keeps what the pullback needs in the workspace (ngpde_edge_layer_workspace_bytes with the
 *           same flag); the workspace must be left untouched until ngpde_edge_layer_backward has run with the SAME descriptor.
 * backward: dy [N][out]; d_state[k] nullable (NULL array: no state gradient); gradients of every weight and bias of phi / update
 *           are written (dbias[l] may be NULL where bias[l] is).
 * Errors: NGPDE_ERR_DIMENSION_MISMATCH when the stacks do not chain or phi's / update's input width is not the message's / the
 * node update's vcat (the reference fails in its matrix product), NGPDE_ERR_WORKSPACE, NGPDE_ERR_UNSUPPORTED for more than four
 * blocks on one side of the message. */
#define NGPDE_MLP_MAX_LAYERS 8
typedef struct ngpde_mlp {
  int32_t n_layers;
  int32_t dims[NGPDE_MLP_MAX_LAYERS + 1];
  int32_t act[NGPDE_MLP_MAX_LAYERS];
  const float *weight[NGPDE_MLP_MAX_LAYERS];
  const float *bias[NGPDE_MLP_MAX_LAYERS];
} ngpde_mlp_t;
typedef struct ngpde_mlp_grad {
  float *dweight[NGPDE_MLP_MAX_LAYERS];
  float *dbias[NGPDE_MLP_MAX_LAYERS];
} ngpde_mlp_grad_t;
typedef enum { NGPDE_LAYER_EDGECONV = 0, NGPDE_LAYER_VMH = 1, NGPDE_LAYER_MPPDE = 2 } ngpde_edge_layer_kind_t;
typedef struct ngpde_edge_layer {
  int32_t kind;                 /* ngpde_edge_layer_kind_t */
  int32_t aggr;                 /* ngpde_aggr_t */
  int32_t n_state;
  const float *state[4];
  int32_t state_width[4];
  const float *node_feat;
  int32_t node_feat_width;
  const float *pos;
  int32_t pos_width;
  const float *edge_feat;
  int32_t edge_feat_width;
  const float *theta;
  int32_t theta_width;
  ngpde_mlp_t phi, update;
} ngpde_edge_layer_t;
size_t ngpde_edge_layer_workspace_bytes(const ngpde_graph_t *g, const ngpde_edge_layer_t *layer, int32_t training);
int32_t ngpde_edge_layer_forward(const ngpde_graph_t *g, const ngpde_edge_layer_t *layer, int32_t training, float *y, void *workspace,
                                 size_t workspace_bytes, ngpde_stream_t stream);
int32_t ngpde_edge_layer_backward(const ngpde_graph_t *g, const ngpde_edge_layer_t *layer, const float *dy, float *const *d_state,
                                  const ngpde_mlp_grad_t *dphi, const ngpde_mlp_grad_t *dupdate, void *workspace,
                                  size_t workspace_bytes, ngpde_stream_t stream);

/* GNOConv (/root/reference/src/layers.jl:509-547) in one call, its pullback in one:
 *   K_e = reshape(phi([s_i; s_j; e_ij]), out, in),  m_i = aggr_j K_e h_j,  y = act.(W h + m + b)
 * h [N][in]; node_feat = vcat(values(g.ndata)...) [N][ds] (:517-519); edge_feat = vcat(values(g.edata)...) [E][de] in p order (:521);
 * phi: dims[0] = 2 ds + de, dims[n_layers] = in * out (column-major reshape: element o + out * i, :527); weight (out x in), bias [out]
 * nullable.  The library picks the form: reassociated (the last Dense of phi without activation: T_j = W2 (x) h_j at node level, the
 * in*out x E tensor is never formed), with the per-edge input formed inside the message launch for a two-layer phi, or the literal
 * batched_mul (NGPDE_GNO_MATERIALIZE=1 forces it).  Workspace / training / pullback conventions as ngpde_edge_layer_*; dh nullable. */
typedef struct ngpde_gno_layer {
  int32_t in_chs, out_chs;
  int32_t aggr;                 /* ngpde_aggr_t */
  int32_t act;                  /* ngpde_act_t of the layer's tail */
  const float *h;
  const float *node_feat;
  int32_t node_feat_width;
  const float *edge_feat;
  int32_t edge_feat_width;
  ngpde_mlp_t phi;
  const float *weight, *bias;
} ngpde_gno_layer_t;
size_t ngpde_gno_layer_workspace_bytes(const ngpde_graph_t *g, const ngpde_gno_layer_t *layer, int32_t training);
int32_t ngpde_gno_layer_forward(const ngpde_graph_t *g, const ngpde_gno_layer_t *layer, int32_t training, float *y, void *workspace,
                                size_t workspace_bytes, ngpde_stream_t stream);
int32_t ngpde_gno_layer_backward(const ngpde_graph_t *g, const ngpde_gno_layer_t *layer, const float *dy, float *dh,
                                 const ngpde_mlp_grad_t *dphi, float *dweight, float *dbias, void *workspace, size_t workspace_bytes,
                                 ngpde_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------------------------
 * Solver level, ONE create call: the device-resident fixed-step neural-ODE plan for a right-hand side, chosen and checked by the library.
 * What a Lux / DiffEqFlux host binds where the tutorials write  NeuralODE(model, tspan, Tsit5(); saveat = ...)  and differentiate through
 * the solve (/root/reference/docs/src/tutorials/graph_node.md:44-66, :78; docs/src/tutorials/VMH.md:85-89, :104-108, :132-141):
 *   rhs = NGPDE_RHS_GCN2: Chain(GCNConv(d => d, act), GCNConv(d => d, act)), width = d in {16, 32, 64, 128}; `members` > 1: a block-diagonal
 *         batch of identical structures on the MEMBER's handle (test/runtests.jl:89-102; ERR_UNSUPPORTED when the persistent plan does not
 *         take it: create again on the batch's own handle with members = 1)            -> ngpde_node_gcn2_create[_batch]
 *   rhs = NGPDE_RHS_GAT:  one GAT-style layer, width = 64 = heads * head_width          -> ngpde_node_gat_create[_batch]
 *   rhs = NGPDE_RHS_VMH:  VMHConv(phi, gamma) on a state of `width` rows per node (1), pos [N][pos_width] (copied), phi / gamma given by
 *         their layer widths and activations; the entry checks that the stacks chain as src/layers.jl:316, :328 feed them
 *         (phi_dims[0] = 2 width + pos_width, gamma_dims[0] = width + phi's output, gamma's output = width: DimensionMismatch otherwise)
 *                                                                                       -> ngpde_node_vmh_create
 * NGPDE_ERR_UNSUPPORTED (text in ngpde_last_error) = no device-resident plan takes this right-hand side on this graph: the host steps the
 * layer itself, every Runge-Kutta combination one ngpde_rk_stage_combine launch (any other right-hand side does that too).
 * *flags (nullable): the NGPDE_NODE_* bits of the plan chosen.  Parameters travel per call (no hidden parameter state):
 *   GCN2: first.weight[0..1], first.bias[0..1] (nullable);  GAT: first.weight[0], attention (2c x heads), first.bias[0] (nullable);
 *   VMH: first = phi's layers, second = gamma's.  Gradients come back in the same places of ngpde_ode_grads_t.
 * forward: out = u(T) [N * members][width], or with save_every > 0 (VMH only) the saved states [T][N] of ngpde_node_vmh_forward_saveat;
 * backward: dout in the same shape, du0 [N * members][width].  One solve in flight per plan (ngpde_node_generation's rule). */
typedef struct ngpde_ode ngpde_ode_t;
typedef enum { NGPDE_RHS_GCN2 = 1, NGPDE_RHS_GAT = 2, NGPDE_RHS_VMH = 3 } ngpde_rhs_t;
typedef struct ngpde_ode_desc {
  int32_t rhs;                 /* ngpde_rhs_t */
  int32_t tableau;             /* ngpde_tableau_t */
  int32_t n_steps, with_backward, members;
  double dt;
  int32_t width;               /* GCN2: d; GAT: 64; VMH: rows of the state per node */
  int32_t act;                 /* GCN2 / GAT: the layers' activation */
  int32_t heads, head_width;   /* GAT */
  float negative_slope;        /* GAT */
  int32_t pos_width, aggr;     /* VMH */
  const float *pos;            /* VMH: device [N][pos_width] */
  int32_t n_phi, phi_dims[NGPDE_MLP_MAX_LAYERS + 1], phi_acts[NGPDE_MLP_MAX_LAYERS];
  int32_t n_gamma, gamma_dims[NGPDE_MLP_MAX_LAYERS + 1], gamma_acts[NGPDE_MLP_MAX_LAYERS];
} ngpde_ode_desc_t;
typedef struct ngpde_ode_wb {
  const float *weight[NGPDE_MLP_MAX_LAYERS];
  const float *bias[NGPDE_MLP_MAX_LAYERS];
} ngpde_ode_wb_t;
typedef struct ngpde_ode_params {
  ngpde_ode_wb_t first, second;
  const float *attention;
} ngpde_ode_params_t;
typedef struct ngpde_ode_grads {
  ngpde_mlp_grad_t first, second;
  float *dattention;
} ngpde_ode_grads_t;
int32_t ngpde_ode_create(const ngpde_graph_t *g, const ngpde_ode_desc_t *desc, ngpde_ode_t **out, int32_t *flags);
int32_t ngpde_ode_destroy(ngpde_ode_t *plan);
size_t ngpde_ode_tape_bytes(const ngpde_ode_t *plan);
int32_t ngpde_ode_fault(ngpde_ode_t *plan, ngpde_stream_t stream, int32_t *fault);
int32_t ngpde_ode_forward(ngpde_ode_t *plan, const float *u0, const ngpde_ode_params_t *params, int32_t save_every, int32_t save_start,
                          float *out, ngpde_stream_t stream);
int32_t ngpde_ode_backward(ngpde_ode_t *plan, const ngpde_ode_params_t *params, int32_t save_every, int32_t save_start, const float *dout,
                           float *du0, const ngpde_ode_grads_t *grads, ngpde_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NGPDE_H */

// api_gcn.hip -- C-ABI entry points of the GCNConv path (declared in include/ngpde.h).
// Replaces (l::GCNConv)(x, ps, st[, edge_weight]) of /root/reference/src/layers.jl:200-239 and the
// pullback Zygote derives for it.
#include <algorithm>

#include "common.h"

using namespace ngpde;

namespace {

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// one block of the Dense kernels' segmented-input table
SegTable one_seg(const float *ptr, int width) {
  SegTable t;
  t.n = 1;
  t.ptr[0] = ptr;
  t.width[0] = width;
  t.vec[0] = (width % 4 == 0 && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0) ? 1 : 0;
  for (int i = 1; i <= 4; ++i) t.offset[i] = width;
  return t;
}
SegGrad one_grad(float *ptr, int width) {
  SegGrad t;
  t.n = 1;
  t.ptr[0] = ptr;
  t.width[0] = width;
  for (int i = 1; i <= 4; ++i) t.offset[i] = width;
  return t;
}

int32_t check_common(const char *fn, const ngpde_graph_t *g, int32_t din, int32_t dout, int32_t act) {
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "%s: graph is NULL", fn);
  NGPDE_REQUIRE(din > 0 && dout > 0, NGPDE_ERR_DIMENSION_MISMATCH, "%s: DimensionMismatch: din=%d dout=%d", fn, din, dout);
  NGPDE_REQUIRE(act >= NGPDE_ACT_IDENTITY && act <= NGPDE_ACT_SOFTPLUS, NGPDE_ERR_INVALID_ARGUMENT,
                "%s: unknown activation code %d", fn, act);
  NGPDE_REQUIRE(g->has_norm, NGPDE_ERR_STATE, "%s: GCN normalisation not set (call ngpde_graph_set_gcn_norm)", fn);
  return NGPDE_OK;
}

}  // namespace

extern "C" {

size_t ngpde_gcn_workspace_bytes(const ngpde_graph_t *g, int32_t din, int32_t dout, int32_t backward) {
  if (!g) return 0;
  const size_t n = (size_t)g->n_nodes;
  if (fused_supported(din, dout)) {
    if (!backward) return 256;
    const size_t nb = (size_t)fused_num_blocks(g->n_nodes);
    return align256(nb * (size_t)din * dout * 4) + align256(nb * (size_t)dout * 4) + align256(n * din * 4) + 256;
  }
  // any-width path: the Dense part on the fp32-MFMA kernels of dense_mfma.hip.  Forward: the aggregated input (dout >= din) or
  // the split-K partial products of W x (dout < din); backward: dz, one [N][max] buffer, the weight-pullback slabs.
  const size_t dmax = (size_t)std::max(din, dout);
  if (!backward) {
    const size_t parts = dout < din ? (size_t)dense_fwd_split_count(din, dense_fwd_splits((int64_t)n, din, dout)) : 1;
    return align256(n * (dout < din ? (size_t)dout * parts : dmax) * 4) + 256;
  }
  // the third region serves two users: the weight pullback's slabs (chunks x (din + 1) x dout) and, when dout < din, the two-stage
  // column sums of the bias gradient (launch_colsum2: kColsumChunks x dout) -- with narrow layers the second is the larger one
  const size_t slabs = (size_t)dense_weight_chunks((int64_t)n, din, dout) * (din + 1) * dout;
  return 2 * align256(n * dmax * 4) + align256(std::max(slabs, (size_t)kColsumChunks * dout) * 4) + 256;
}

int32_t ngpde_gcn_forward(const ngpde_graph_t *g, int32_t din, int32_t dout, int32_t act, const float *x,
                          const float *weight, const float *bias, float *y, float *save_agg, float *save_z,
                          void *workspace, size_t workspace_bytes, ngpde_stream_t stream_) {
  NGPDE_RANGE();
  int32_t st = check_common("ngpde_gcn_forward", g, din, dout, act);
  if (st) return st;
  if (g->n_nodes == 0) return NGPDE_OK;  // EMPTYGRAPH (src/layers.jl:14): zero columns in, zero columns out
  NGPDE_REQUIRE(x && weight && y, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gcn_forward: x/weight/y is NULL");
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t n = g->n_nodes;
  if (fused_supported(din, dout)) {
    FusedFwdArgs a;
    a.g = g; a.d = din; a.act = act; a.x = x; a.wt = weight; a.bias = bias; a.y = y;
    a.save_agg = save_agg; a.save_z = save_z;
    return launch_fused_fwd(a, stream);
  }
  const size_t need = ngpde_gcn_workspace_bytes(g, din, dout, 0);
  NGPDE_REQUIRE(workspace && workspace_bytes >= need, NGPDE_ERR_WORKSPACE,
                "ngpde_gcn_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
  float *tmp = (float *)workspace;
  if (dout >= din) {  // aggregate, then multiply (src/layers.jl:235-237)
    float *agg = save_agg ? save_agg : tmp;
    if ((st = launch_spmm_generic(g, false, true, din, NGPDE_AGGR_SUM, x, nullptr, agg, stream))) return st;
    return launch_dense_seg_fwd(n, one_seg(agg, din), din, dout, act, weight, bias, y, save_z, stream);
  }
  // multiply first (src/layers.jl:220-223) -- split over the input features when the row tiles alone cannot fill the chip,
  // the partial products summed in slab order --, then ONE launch: aggregation + bias + activation
  const int nsplit = dense_fwd_splits(n, din, dout);
  if ((st = launch_dense_seg_fwd_splitk(n, one_seg(x, din), din, dout, weight, tmp, nsplit, stream))) return st;
  if ((st = launch_sum_partials(n * dout, dense_fwd_split_count(din, nsplit), (size_t)n * dout, tmp, stream))) return st;
  return launch_spmm_gcn_tail(g, dout, tmp, bias, act, y, save_z, stream);
}

}  // extern "C"

namespace {
// extra bytes behind ngpde_gcn_workspace_bytes(.., 1) when the gradient w.r.t. the edge_weight argument is asked for: dx (the
// caller may pass none), x W (Dout < Din: the array that entered the propagation), the per-node degree terms
size_t ew_extra_bytes(const ngpde_graph_t *g, int32_t din, int32_t dout) {
  const size_t n = (size_t)g->n_nodes;
  return align256(n * din * 4) + align256(n * (size_t)std::min(din, dout) * 4) + align256(n * 4);
}

// bias: only read when dedge_weight is asked for and Dout < Din (x3 = z - bias there)
int32_t gcn_backward_impl(const char *fn, const ngpde_graph_t *g, int32_t din, int32_t dout, int32_t act, const float *x, const float *weight,
                          const float *bias, const float *z, const float *saved_agg, const float *dy, float *dx, float *dweight, float *dbias,
                          float *dedge_weight, void *workspace, size_t workspace_bytes, ngpde_stream_t stream_) {
  int32_t st = check_common(fn, g, din, dout, act);
  if (st) return st;
  NGPDE_REQUIRE(dweight != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gcn_backward: dweight is NULL");
  if (dedge_weight && g->n_edges > 0 && g->n_nodes == 0) return fail(NGPDE_ERR_STATE, "edges without nodes");
  if (g->n_nodes == 0) {  // empty graph: zero gradients
    { const int32_t zs = launch_zero(dweight, (size_t)din * dout * 4, (hipStream_t)stream_); if (zs) return zs; }
    if (dbias) { const int32_t zs = launch_zero(dbias, (size_t)dout * 4, (hipStream_t)stream_); if (zs) return zs; }
    return NGPDE_OK;
  }
  NGPDE_REQUIRE(weight && z && dy, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gcn_backward: weight/z/dy is NULL");
  NGPDE_REQUIRE(dout < din || saved_agg, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gcn_backward: saved_agg is NULL");
  NGPDE_REQUIRE(dout >= din || x, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gcn_backward: x is NULL");
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t n = g->n_nodes;
  const size_t base = ngpde_gcn_workspace_bytes(g, din, dout, 1);
  const size_t need = base + (dedge_weight ? ew_extra_bytes(g, din, dout) : 0);
  NGPDE_REQUIRE(workspace && workspace_bytes >= need, NGPDE_ERR_WORKSPACE,
                "ngpde_gcn_backward: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
  char *ws = (char *)workspace;
  float *ew_dx = nullptr, *ew_xw = nullptr, *ew_nd = nullptr;
  if (dedge_weight) {
    NGPDE_REQUIRE(x != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gcn_backward_ew: x is NULL");
    ew_dx = (float *)(ws + base);
    ew_xw = (float *)(ws + base + align256((size_t)n * din * 4));
    ew_nd = (float *)(ws + base + align256((size_t)n * din * 4) + align256((size_t)n * (size_t)std::min(din, dout) * 4));
    if (!dx) dx = ew_dx;      // the degree term needs the layer's input gradient
  }
  if (fused_supported(din, dout)) {
    const size_t nb = (size_t)fused_num_blocks(n);
    float *slab_dw = (float *)ws;
    float *slab_db = (float *)(ws + align256(nb * (size_t)din * dout * 4));
    float *gbuf = (float *)((char *)slab_db + align256(nb * (size_t)dout * 4));
    { const int32_t zs = launch_zero(slab_dw, (size_t)((char *)gbuf - ws), stream); if (zs) return zs; }
    FusedBwdArgs a;
    a.g = g; a.d = din; a.act = act; a.aggregate = false; a.g_in = dy;
    a.do_dense = true; a.z = z; a.saved_agg = saved_agg; a.wt = weight; a.g_out = gbuf;
    a.slab_dw = slab_dw; a.slab_db = slab_db;
    if ((st = launch_fused_bwd(a, stream))) return st;
    if (dx) {
      FusedBwdArgs b;
      b.g = g; b.d = din; b.act = act; b.aggregate = true; b.g_in = gbuf; b.do_dense = false; b.store_t = dx;
      if ((st = launch_fused_bwd(b, stream))) return st;
    }
    const int ns = fused_num_slabs(n, din);
    if ((st = launch_reduce_slabs(slab_dw, ns, din * dout, din / 16, dweight, stream))) return st;
    if (dbias && (st = launch_reduce_slabs(slab_db, ns, dout, 0, dbias, stream))) return st;
    // gbuf = dz W^T = dL/dx3, saved_agg = x3, the propagated array is x itself
    if (dedge_weight) return launch_gcn_edge_weight_grad(g, din, gbuf, saved_agg, nullptr, dx, x, ew_nd, dedge_weight, stream);
    return NGPDE_OK;
  }
  const size_t dmax = (size_t)std::max(din, dout);
  float *dz = (float *)ws;
  float *tmp = (float *)(ws + align256((size_t)n * dmax * 4));
  float *partial = (float *)(ws + 2 * align256((size_t)n * dmax * 4));
  if ((st = launch_act_bwd(n * dout, act, dy, z, dz, stream))) return st;
  if (dout >= din) {   // y = act(agg W + b): dW = agg^T dz, db = column sums of dz (the weight pullback's bias row), dx = A^T (dz W^T)
    if ((st = launch_dense_seg_bwd_weight(n, one_seg(saved_agg, din), din, dout, dz, dweight, dbias, partial, stream))) return st;
    if (dx) {
      if ((st = launch_dense_seg_bwd_input(n, one_grad(tmp, din), din, dout, dz, weight, stream))) return st;
      if ((st = launch_spmm_generic(g, true, true, din, NGPDE_AGGR_SUM, tmp, nullptr, dx, stream))) return st;
      // tmp = dz W^T = dL/dx3, saved_agg = x3, the propagated array is x itself
      if (dedge_weight) return launch_gcn_edge_weight_grad(g, din, tmp, saved_agg, nullptr, dx, x, ew_nd, dedge_weight, stream);
    }
    return NGPDE_OK;
  }
  // y = act(A (x W) + b): db from dz; g = A^T dz; dW = x^T g; dx = g W^T
  if (dbias && (st = launch_colsum2(n, dout, dz, partial, dbias, stream))) return st;
  if ((st = launch_spmm_generic(g, true, true, dout, NGPDE_AGGR_SUM, dz, nullptr, tmp, stream))) return st;
  if ((st = launch_dense_seg_bwd_weight(n, one_seg(x, din), din, dout, tmp, dweight, nullptr, partial, stream))) return st;
  if (dx && (st = launch_dense_seg_bwd_input(n, one_grad(dx, din), din, dout, tmp, weight, stream))) return st;
  if (dedge_weight) {
    // the propagated array was x W (recomputed), its gradient is tmp = A^T dz, dL/dx3 = dz and x3 = z - bias
    if ((st = launch_dense_seg_fwd(n, one_seg(x, din), din, dout, NGPDE_ACT_IDENTITY, weight, nullptr, ew_xw, nullptr, stream))) return st;
    return launch_gcn_edge_weight_grad(g, dout, dz, z, bias, tmp, ew_xw, ew_nd, dedge_weight, stream);
  }
  return NGPDE_OK;
}
}  // namespace

extern "C" {

int32_t ngpde_gcn_backward(const ngpde_graph_t *g, int32_t din, int32_t dout, int32_t act, const float *x,
                           const float *weight, const float *z, const float *saved_agg, const float *dy, float *dx,
                           float *dweight, float *dbias, void *workspace, size_t workspace_bytes,
                           ngpde_stream_t stream_) {
  NGPDE_RANGE();
  return gcn_backward_impl("ngpde_gcn_backward", g, din, dout, act, x, weight, nullptr, z, saved_agg, dy, dx, dweight, dbias, nullptr, workspace,
                           workspace_bytes, stream_);
}

size_t ngpde_gcn_backward_ew_workspace_bytes(const ngpde_graph_t *g, int32_t din, int32_t dout) {
  if (!g) return 0;
  return ngpde_gcn_workspace_bytes(g, din, dout, 1) + ew_extra_bytes(g, din, dout);
}

int32_t ngpde_gcn_backward_ew(const ngpde_graph_t *g, int32_t din, int32_t dout, int32_t act, const float *x, const float *weight,
                              const float *bias, const float *z, const float *saved_agg, const float *dy, float *dx, float *dweight,
                              float *dbias, float *dedge_weight, void *workspace, size_t workspace_bytes, ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(dedge_weight != nullptr || (g && g->n_edges == 0), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gcn_backward_ew: dedge_weight is NULL");
  return gcn_backward_impl("ngpde_gcn_backward_ew", g, din, dout, act, x, weight, bias, z, saved_agg, dy, dx, dweight, dbias, dedge_weight, workspace,
                           workspace_bytes, stream_);
}

int32_t ngpde_propagate_copy_xj(const ngpde_graph_t *g, int32_t d, int32_t aggr, int32_t by_source, const float *x,
                                const float *edge_weight, float *out, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_propagate_copy_xj: graph is NULL");
  NGPDE_REQUIRE(d > 0 && x && out, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_propagate_copy_xj: bad arguments");
  return launch_spmm_generic(g, by_source != 0, false, d, aggr, x, edge_weight, out, (hipStream_t)stream);
}

}  // extern "C"

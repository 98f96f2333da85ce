// api_mp.hip -- C-ABI entry points of the message-passing primitives (declared in include/ngpde.h).
// Together they replace the bodies of the reference's edge-function layers:
//   propagate(message, g, aggr; xi, xj, e)   /root/reference/src/layers.jl:111, :326, :416, :534, :656
#include <algorithm>

#include "common.h"

using namespace ngpde;

namespace {

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// n_rows = 0: a Dense over no rows (an edgeless graph's message MLP: /root/reference's scatter over an empty edge set gives zeros) --
// its blocks may be NULL, as the empty arrays of most hosts are
int32_t make_segs(const char *fn, int32_t n_seg, const float *const *seg_ptr, const int32_t *seg_width,
                  const int32_t *seg_row_div, SegTable &t, int *din, int64_t n_rows = -1) {
  NGPDE_REQUIRE(n_seg >= 1 && n_seg <= 4, NGPDE_ERR_INVALID_ARGUMENT, "%s: 1..4 input blocks supported, got %d", fn, n_seg);
  NGPDE_REQUIRE(seg_ptr && seg_width, NGPDE_ERR_INVALID_ARGUMENT, "%s: NULL block table", fn);
  t.n = n_seg;
  int off = 0;
  for (int i = 0; i < n_seg; ++i) {
    NGPDE_REQUIRE(seg_width[i] >= 0, NGPDE_ERR_DIMENSION_MISMATCH, "%s: negative block width", fn);
    NGPDE_REQUIRE(seg_width[i] == 0 || n_rows == 0 || seg_ptr[i], NGPDE_ERR_INVALID_ARGUMENT, "%s: block %d is NULL", fn, i);
    t.ptr[i] = seg_ptr[i];
    t.width[i] = seg_width[i];
    t.row_div[i] = (seg_row_div && seg_row_div[i] > 0) ? seg_row_div[i] : 1;
    t.offset[i] = off;
    t.vec[i] = (off % 4 == 0 && seg_width[i] % 4 == 0 && (reinterpret_cast<uintptr_t>(seg_ptr[i]) & 15) == 0) ? 1 : 0;
    off += seg_width[i];
  }
  for (int i = n_seg; i <= 4; ++i) t.offset[i] = off;
  *din = off;
  return NGPDE_OK;
}

int32_t check_act(const char *fn, int32_t act) {
  NGPDE_REQUIRE(act >= NGPDE_ACT_IDENTITY && act <= NGPDE_ACT_SOFTPLUS, NGPDE_ERR_INVALID_ARGUMENT,
                "%s: unknown activation code %d", fn, act);
  return NGPDE_OK;
}

}  // namespace

extern "C" {

int32_t ngpde_dense_forward(int64_t n, int32_t n_seg, const float *const *seg_ptr, const int32_t *seg_width,
                            const int32_t *seg_row_div, int32_t dout, int32_t act, const float *weight,
                            const float *bias, float *y, float *save_z, ngpde_stream_t stream) {
  NGPDE_RANGE();
  SegTable t;
  int din = 0;
  int32_t st = make_segs("ngpde_dense_forward", n_seg, seg_ptr, seg_width, seg_row_div, t, &din, n);
  if (st || (st = check_act("ngpde_dense_forward", act))) return st;
  NGPDE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && dout > 0, NGPDE_ERR_DIMENSION_MISMATCH,
                "ngpde_dense_forward: DimensionMismatch (rows must be in [0, 2^31), dout > 0)");
  if (n == 0) return NGPDE_OK;
  NGPDE_REQUIRE(weight && y, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_dense_forward: weight/y is NULL");
  return launch_dense_seg_fwd(n, t, din, dout, act, weight, bias, y, save_z, (hipStream_t)stream);
}

int32_t ngpde_dense_multi_forward(int32_t count, const int64_t *n, const int32_t *n_seg, const float *const *seg_ptr,
                                  const int32_t *seg_width, const int32_t *seg_row_div, const int32_t *dout, const int32_t *act,
                                  const float *const *weight, const float *const *bias, float *const *y, float *const *save_z,
                                  ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(count >= 1 && count <= 4, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_dense_multi_forward: 1..4 problems (got %d)", count);
  NGPDE_REQUIRE(n && n_seg && seg_ptr && seg_width && dout && act && weight && bias && y && save_z, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_dense_multi_forward: NULL table");
  SegTable t[4];
  int din[4], dd[4], aa[4];
  int off = 0;
  for (int q = 0; q < count; ++q) {
    int32_t st = make_segs("ngpde_dense_multi_forward", n_seg[q], seg_ptr + off, seg_width + off, seg_row_div ? seg_row_div + off : nullptr,
                           t[q], &din[q], n[q]);
    if (st || (st = check_act("ngpde_dense_multi_forward", act[q]))) return st;
    NGPDE_REQUIRE(n[q] >= 0 && n[q] < ((int64_t)1 << 31) && dout[q] > 0, NGPDE_ERR_DIMENSION_MISMATCH,
                  "ngpde_dense_multi_forward: DimensionMismatch in problem %d (rows must be in [0, 2^31), dout > 0)", q);
    NGPDE_REQUIRE(n[q] == 0 || (weight[q] && y[q]), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_dense_multi_forward: weight/y of problem %d is NULL", q);
    dd[q] = dout[q]; aa[q] = act[q];
    off += n_seg[q];
  }
  return launch_dense_multi_fwd(count, n, t, din, dd, aa, weight, bias, y, save_z, (hipStream_t)stream);
}

int32_t ngpde_dense_pair_forward(int64_t n, int32_t n_seg_a, const float *const *seg_ptr_a, const int32_t *seg_width_a,
                                 const int32_t *seg_row_div_a, int32_t dout_a, int32_t act_a, const float *weight_a, const float *bias_a,
                                 float *y_a, float *save_z_a, int32_t n_seg_b, const float *const *seg_ptr_b,
                                 const int32_t *seg_width_b, const int32_t *seg_row_div_b, int32_t dout_b, int32_t act_b,
                                 const float *weight_b, const float *bias_b, float *y_b, float *save_z_b, ngpde_stream_t stream) {
  NGPDE_RANGE();
  SegTable ta, tb;
  int dina = 0, dinb = 0;
  int32_t st = make_segs("ngpde_dense_pair_forward", n_seg_a, seg_ptr_a, seg_width_a, seg_row_div_a, ta, &dina);
  if (st || (st = make_segs("ngpde_dense_pair_forward", n_seg_b, seg_ptr_b, seg_width_b, seg_row_div_b, tb, &dinb))) return st;
  if ((st = check_act("ngpde_dense_pair_forward", act_a)) || (st = check_act("ngpde_dense_pair_forward", act_b))) return st;
  NGPDE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && dout_a > 0 && dout_b > 0, NGPDE_ERR_DIMENSION_MISMATCH,
                "ngpde_dense_pair_forward: DimensionMismatch (rows must be in [0, 2^31), dout > 0)");
  if (n == 0) return NGPDE_OK;
  NGPDE_REQUIRE(weight_a && y_a && weight_b && y_b, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_dense_pair_forward: weight/y is NULL");
  if (dense_pair_fwd_applicable(n, ta, dina, dout_a, tb, dinb, dout_b))
    return launch_dense_pair_fwd(n, ta, dina, dout_a, act_a, weight_a, bias_a, y_a, save_z_a, tb, dinb, dout_b, act_b, weight_b, bias_b,
                                 y_b, save_z_b, (hipStream_t)stream);
  if ((st = launch_dense_seg_fwd(n, ta, dina, dout_a, act_a, weight_a, bias_a, y_a, save_z_a, (hipStream_t)stream))) return st;
  return launch_dense_seg_fwd(n, tb, dinb, dout_b, act_b, weight_b, bias_b, y_b, save_z_b, (hipStream_t)stream);
}

int32_t ngpde_dense_chain2_fused(int64_t n, int32_t n_seg, const float *const *seg_ptr, const int32_t *seg_width,
                                 const int32_t *seg_row_div, int32_t dmid, int32_t dout) {
  SegTable t;
  int din = 0;
  if (make_segs("ngpde_dense_chain2_fused", n_seg, seg_ptr, seg_width, seg_row_div, t, &din)) return 0;
  return (n > 0 && n < ((int64_t)1 << 31) && dense_chain_fwd_applicable(n, t, din, dmid, dout)) ? 1 : 0;
}

int32_t ngpde_dense_chain2_forward(int64_t n, int32_t n_seg, const float *const *seg_ptr, const int32_t *seg_width,
                                   const int32_t *seg_row_div, int32_t dmid, int32_t act1, const float *weight1, const float *bias1,
                                   float *a1, float *save_z1, int32_t dout, int32_t act2, const float *weight2, const float *bias2,
                                   float *y, float *save_z2, ngpde_stream_t stream) {
  NGPDE_RANGE();
  SegTable t;
  int din = 0;
  int32_t st = make_segs("ngpde_dense_chain2_forward", n_seg, seg_ptr, seg_width, seg_row_div, t, &din);
  if (st || (st = check_act("ngpde_dense_chain2_forward", act1)) || (st = check_act("ngpde_dense_chain2_forward", act2))) return st;
  NGPDE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && dmid > 0 && dout > 0, NGPDE_ERR_DIMENSION_MISMATCH,
                "ngpde_dense_chain2_forward: DimensionMismatch (rows must be in [0, 2^31), widths > 0)");
  if (n == 0) return NGPDE_OK;
  NGPDE_REQUIRE(weight1 && weight2 && y, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_dense_chain2_forward: weight/y is NULL");
  if (dense_chain_fwd_applicable(n, t, din, dmid, dout))
    return launch_dense_chain_fwd(n, t, din, act1, weight1, bias1, a1, save_z1, dout, act2, weight2, bias2, y, save_z2, (hipStream_t)stream);
  NGPDE_REQUIRE(a1 != nullptr, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_dense_chain2_forward: this shape runs as two launches and needs the [n][dmid] buffer a1 (ngpde_dense_chain2_fused)");
  if ((st = launch_dense_seg_fwd(n, t, din, dmid, act1, weight1, bias1, a1, save_z1, (hipStream_t)stream))) return st;
  SegTable t2;
  const float *p2[1] = {a1};
  const int32_t w2[1] = {dmid};
  int din2 = 0;
  if ((st = make_segs("ngpde_dense_chain2_forward", 1, p2, w2, nullptr, t2, &din2))) return st;
  return launch_dense_seg_fwd(n, t2, din2, dout, act2, weight2, bias2, y, save_z2, (hipStream_t)stream);
}

size_t ngpde_dense_pair_backward_workspace_bytes(int64_t n, int32_t n_seg_a, const float *const *seg_ptr_a, const int32_t *seg_width_a,
                                                 const int32_t *seg_row_div_a, int32_t n_seg_b, const float *const *seg_ptr_b,
                                                 const int32_t *seg_width_b, const int32_t *seg_row_div_b, int32_t dout) {
  SegTable ta, tb;
  int dina = 0, dinb = 0;
  if (make_segs("ngpde_dense_pair_backward", n_seg_a, seg_ptr_a, seg_width_a, seg_row_div_a, ta, &dina) ||
      make_segs("ngpde_dense_pair_backward", n_seg_b, seg_ptr_b, seg_width_b, seg_row_div_b, tb, &dinb) || dout != 64)
    return 0;
  const int grid = dense_pair_bwd_grid(n, ta, dina, tb, dinb);
  return grid ? dense_pair_bwd_workspace(grid, dina, dinb) : 0;
}

int32_t ngpde_dense_pair_backward(int64_t n, int32_t n_seg_a, const float *const *seg_ptr_a, const int32_t *seg_width_a,
                                  const int32_t *seg_row_div_a, const float *weight_a, const float *dy_a, float *dweight_a,
                                  float *dbias_a, int32_t n_seg_b, const float *const *seg_ptr_b, const int32_t *seg_width_b,
                                  const int32_t *seg_row_div_b, const float *weight_b, const float *dy_b, float *dweight_b,
                                  float *dbias_b, int32_t dout, float *dx, const float *dx_addend, void *workspace,
                                  size_t workspace_bytes, ngpde_stream_t stream) {
  NGPDE_RANGE();
  SegTable ta, tb;
  int dina = 0, dinb = 0;
  int32_t st = make_segs("ngpde_dense_pair_backward", n_seg_a, seg_ptr_a, seg_width_a, seg_row_div_a, ta, &dina);
  if (st || (st = make_segs("ngpde_dense_pair_backward", n_seg_b, seg_ptr_b, seg_width_b, seg_row_div_b, tb, &dinb))) return st;
  const int grid = (dout == 64 && n > 0) ? dense_pair_bwd_grid(n, ta, dina, tb, dinb) : 0;
  NGPDE_REQUIRE(grid > 0, NGPDE_ERR_UNSUPPORTED,
                "ngpde_dense_pair_backward: needs 64 outputs, a shared 16-byte aligned 64-wide leading block, <= 4 narrow features per "
                "side and >= 32768 rows (ngpde_dense_pair_backward_workspace_bytes returns 0 otherwise): use two ngpde_dense_backward calls");
  NGPDE_REQUIRE(weight_a && weight_b && dy_a && dy_b && dweight_a && dweight_b && dx, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_dense_pair_backward: NULL argument");
  NGPDE_REQUIRE(workspace && workspace_bytes >= dense_pair_bwd_workspace(grid, dina, dinb), NGPDE_ERR_WORKSPACE,
                "ngpde_dense_pair_backward: workspace too small");
  return launch_dense_pair_bwd(n, ta, dina, weight_a, dy_a, dweight_a, dbias_a, tb, dinb, weight_b, dy_b, dweight_b, dbias_b, dx, dx_addend,
                               workspace, grid, (hipStream_t)stream);
}

size_t ngpde_dense_workspace_bytes(int64_t n, int32_t din_total, int32_t dout) {
  return align256((size_t)std::max<int64_t>(n, 1) * dout * 4) +
         align256((size_t)dense_weight_chunks(n, din_total, dout) * (din_total + 1) * dout * 4) +
         align256(dense_bwd_input_split_bytes(n, din_total, dout)) + 256;
}

int32_t ngpde_dense_backward(int64_t n, int32_t n_seg, const float *const *seg_ptr, const int32_t *seg_width,
                             const int32_t *seg_row_div, int32_t dout, int32_t act, const float *weight, const float *z,
                             const float *dy, float *const *dseg_ptr, float *dweight, float *dbias, void *workspace,
                             size_t workspace_bytes, ngpde_stream_t stream_) {
  NGPDE_RANGE();
  SegTable t;
  int din = 0;
  int32_t st = make_segs("ngpde_dense_backward", n_seg, seg_ptr, seg_width, seg_row_div, t, &din, n);
  if (st || (st = check_act("ngpde_dense_backward", act))) return st;
  hipStream_t stream = (hipStream_t)stream_;
  NGPDE_REQUIRE(dweight != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_dense_backward: dweight is NULL");
  NGPDE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), NGPDE_ERR_DIMENSION_MISMATCH, "ngpde_dense_backward: rows must be in [0, 2^31)");
  if (n == 0) {
    { const int32_t zs = launch_zero(dweight, (size_t)din * dout * 4, stream); if (zs) return zs; }
    if (dbias) { const int32_t zs = launch_zero(dbias, (size_t)dout * 4, stream); if (zs) return zs; }
    return NGPDE_OK;
  }
  NGPDE_REQUIRE(weight && dy && (z || act == NGPDE_ACT_IDENTITY), NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_dense_backward: weight/z/dy is NULL");
  const size_t need = ngpde_dense_workspace_bytes(n, din, dout);
  NGPDE_REQUIRE(workspace && workspace_bytes >= need, NGPDE_ERR_WORKSPACE,
                "ngpde_dense_backward: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
  if (const int sgrid = dense_stream_bwd_grid(n, t, din, dout, dseg_ptr))   // one streaming launch (slabs in the dz area)
    return launch_dense_stream_bwd(n, t, din, act, weight, z, dy, dseg_ptr, dweight, dbias, (float *)workspace, sgrid, stream);
  if (const int sgrid = dense_small_bwd_grid(n, din, dout))                   // at most 64 x 64, latency-bound row counts: one launch too
    return launch_dense_small_bwd(n, t, din, dout, act, weight, z, dy, dseg_ptr, dweight, dbias, (float *)workspace, sgrid, stream);
  float *dz = (float *)workspace;
  float *partial = (float *)((char *)workspace + align256((size_t)n * dout * 4));
  if (act == NGPDE_ACT_IDENTITY) {
    dz = const_cast<float *>(dy);   // read-only below
  } else if ((st = launch_dense_dz(n * dout, act, dy, z, dz, stream))) {
    return st;
  }
  if ((st = launch_dense_seg_bwd_weight(n, t, din, dout, dz, dweight, dbias, partial, stream))) return st;
  if (dseg_ptr) {
    SegGrad gsg;
    gsg.n = t.n;
    bool any = false;
    for (int i = 0; i < t.n; ++i) {
      gsg.ptr[i] = (t.row_div[i] == 1) ? dseg_ptr[i] : nullptr;   // per-graph blocks carry no gradient (@ignore_derivatives, :397,:418)
      gsg.width[i] = t.width[i];
      any = any || gsg.ptr[i];
    }
    for (int i = 0; i <= 4; ++i) gsg.offset[i] = t.offset[i];
    if (any && t.n == 1 && dense_bwd_input_splits(n, din, dout) > 1) {   // few rows, very wide output: split the contraction
      float *split_part = (float *)((char *)partial + align256((size_t)dense_weight_chunks(n, din, dout) * (din + 1) * dout * 4));
      if ((st = launch_dense_bwd_input_splitk(n, gsg.ptr[0], din, dout, dz, weight, split_part, stream))) return st;
    } else if (any && (st = launch_dense_seg_bwd_input(n, gsg, din, dout, dz, weight, stream))) return st;
  }
  return NGPDE_OK;
}

int32_t ngpde_edge_permute(const ngpde_graph_t *g, int32_t d, int32_t inverse, const float *src, float *dst,
                           ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_permute: graph is NULL");
  if (g->n_edges == 0 || d == 0) return NGPDE_OK;
  NGPDE_REQUIRE(src && dst && d > 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_permute: bad arguments");
  return launch_edge_permute(g, d, inverse != 0, src, dst, (hipStream_t)stream);
}

int32_t ngpde_edge_combine_forward(const ngpde_graph_t *g, int32_t h, int32_t act, const float *p_target,
                                   const float *q_source, const float *e_term, float *a_out, float *z_out,
                                   ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_combine_forward: graph is NULL");
  int32_t st = check_act("ngpde_edge_combine_forward", act);
  if (st) return st;
  if (g->n_edges == 0) return NGPDE_OK;
  NGPDE_REQUIRE(h > 0 && a_out, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_combine_forward: bad arguments");
  return launch_edge_combine_fwd(g, h, act, p_target, q_source, e_term, a_out, z_out, (hipStream_t)stream);
}

int32_t ngpde_edge_combine_backward(const ngpde_graph_t *g, int32_t h, int32_t act, const float *da, const float *z,
                                    float *dz, float *dp_target, float *dq_source, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_combine_backward: graph is NULL");
  int32_t st = check_act("ngpde_edge_combine_backward", act);
  if (st) return st;
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(h > 0 && (g->n_edges == 0 || (da && dz)), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_combine_backward: bad arguments");
  return launch_edge_combine_bwd(g, h, act, da, z, dz, dp_target, dq_source, (hipStream_t)stream);
}

int32_t ngpde_segment_reduce_forward(const ngpde_graph_t *g, int32_t d, int32_t aggr, const float *m, float *out,
                                     ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_segment_reduce_forward: graph is NULL");
  NGPDE_REQUIRE(aggr >= NGPDE_AGGR_SUM && aggr <= NGPDE_AGGR_MUL, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_segment_reduce_forward: unknown aggregation %d", aggr);
  if (g->n_nodes == 0 || d == 0) return NGPDE_OK;
  NGPDE_REQUIRE(out && (m || g->n_edges == 0), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_segment_reduce_forward: NULL argument");
  return launch_segment_reduce_fwd(g, d, aggr, m, out, (hipStream_t)stream);
}

int32_t ngpde_segment_reduce_backward(const ngpde_graph_t *g, int32_t d, int32_t aggr, const float *m, const float *out,
                                      const float *dout, float *dm, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_segment_reduce_backward: graph is NULL");
  NGPDE_REQUIRE(aggr >= NGPDE_AGGR_SUM && aggr <= NGPDE_AGGR_MUL, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_segment_reduce_backward: unknown aggregation %d", aggr);
  if (g->n_edges == 0 || d == 0) return NGPDE_OK;
  NGPDE_REQUIRE(dout && dm, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_segment_reduce_backward: NULL argument");
  return launch_segment_reduce_bwd(g, d, aggr, m, out, dout, dm, (hipStream_t)stream);
}

int32_t ngpde_gno_contract_forward(const ngpde_graph_t *g, int32_t cin, int32_t cout, const float *k, const float *h,
                                   float *m, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_contract_forward: graph is NULL");
  if (g->n_edges == 0) return NGPDE_OK;
  NGPDE_REQUIRE(cin > 0 && cout > 0 && k && h && m, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_contract_forward: bad arguments");
  return launch_gno_contract_fwd(g, cin, cout, k, h, m, (hipStream_t)stream);
}

int32_t ngpde_gno_contract_backward(const ngpde_graph_t *g, int32_t cin, int32_t cout, const float *k, const float *h,
                                    const float *dm, float *dk, float *dh, void *workspace, size_t workspace_bytes,
                                    ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_contract_backward: graph is NULL");
  hipStream_t stream = (hipStream_t)stream_;
  if (g->n_edges == 0) {
    if (dh && g->n_nodes) { const int32_t zs = launch_zero(dh, (size_t)g->n_nodes * cin * 4, stream); if (zs) return zs; }
    return NGPDE_OK;
  }
  NGPDE_REQUIRE(cin > 0 && cout > 0 && k && h && dm, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_contract_backward: bad arguments");
  const size_t need = (size_t)g->n_edges * cin * 4;
  NGPDE_REQUIRE(!dh || (workspace && workspace_bytes >= need), NGPDE_ERR_WORKSPACE,
                "ngpde_gno_contract_backward: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
  int32_t st = launch_gno_contract_bwd(g, cin, cout, k, h, dm, dk, dh ? (float *)workspace : nullptr, stream);
  if (st) return st;
  if (dh) return launch_edge_sum_by_source(g, cin, (const float *)workspace, dh, stream);
  return NGPDE_OK;
}

int32_t ngpde_gno_apply_supported(int32_t cout, int32_t kdim) { return gno_apply_supported(cout, kdim) ? 1 : 0; }

int32_t ngpde_gno_apply_forward(const ngpde_graph_t *g, int32_t cout, int32_t kdim, const float *t, const float *bh,
                                const float *z, float *m, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_apply_forward: graph is NULL");
  NGPDE_REQUIRE(gno_apply_supported(cout, kdim), NGPDE_ERR_UNSUPPORTED,
                "ngpde_gno_apply_forward: out = %d, k = %d outside the reassociated path (out, k <= 256; T_j [k][out] must fit 60 KB of LDS)", cout, kdim);
  if (g->n_edges == 0) return NGPDE_OK;
  NGPDE_REQUIRE(t && z && m, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_apply_forward: NULL argument");
  return launch_gno_apply_fwd(g, cout, kdim, t, bh, z, m, (hipStream_t)stream);
}

int32_t ngpde_gno_apply_backward(const ngpde_graph_t *g, int32_t cout, int32_t kdim, const float *t, const float *z,
                                 const float *dm, float *dt, float *dbh, float *dz, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_apply_backward: graph is NULL");
  NGPDE_REQUIRE(gno_apply_supported(cout, kdim), NGPDE_ERR_UNSUPPORTED, "ngpde_gno_apply_backward: unsupported out = %d, k = %d", cout, kdim);
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(t && (g->n_edges == 0 || (z && dm)), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_apply_backward: NULL argument");
  return launch_gno_apply_bwd(g, cout, kdim, t, z, dm, dt, dbh, dz, (hipStream_t)stream);
}

int32_t ngpde_gno_message_backward_from_nodes(const ngpde_graph_t *g, int32_t cout, int32_t kdim, int32_t aggr, int32_t act1,
                                              const float *t, const float *z, const float *dagg, float *dt, float *dbh, float *dz,
                                              float *dq, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_message_backward_from_nodes: graph is NULL");
  NGPDE_REQUIRE(act1 == NGPDE_ACT_IDENTITY || act1 == NGPDE_ACT_RELU, NGPDE_ERR_UNSUPPORTED,
                "ngpde_gno_message_backward_from_nodes: act1 must be identity or relu (the activated input stands for the "
                "pre-activation), got %d", act1);
  NGPDE_REQUIRE(aggr == NGPDE_AGGR_SUM || aggr == NGPDE_AGGR_MEAN, NGPDE_ERR_UNSUPPORTED,
                "ngpde_gno_message_backward_from_nodes: aggregation %d has no node-level form (sum and mean do); use "
                "ngpde_segment_reduce_backward + ngpde_gno_apply_backward", aggr);
  NGPDE_REQUIRE(gno_apply_mfma_supported(cout, kdim), NGPDE_ERR_UNSUPPORTED,
                "ngpde_gno_message_backward_from_nodes: needs out a multiple of 16 (<= 256) and k in {16, 32, 64}, got out = %d, k = %d",
                cout, kdim);
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(t && (g->n_edges == 0 || (z && dagg)) && (dz || !dq), NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_gno_message_backward_from_nodes: NULL argument (dq needs dz)");
  return launch_gno_apply_mfma_bwd(g, cout, kdim, t, z, nullptr, dt, dbh, dz, (hipStream_t)stream, dagg, aggr == NGPDE_AGGR_MEAN ? 1 : 0,
                                   act1, dq);
}

int32_t ngpde_gno_message_supported(int32_t cout, int32_t kdim) { return gno_apply_mfma_supported(cout, kdim) ? 1 : 0; }

int32_t ngpde_gno_message_forward(const ngpde_graph_t *g, int32_t cout, int32_t kdim, int32_t act1, const float *p_target,
                                  const float *q_source, const float *e_term, const float *t, const float *bh, float *z_out, float *m,
                                  ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_message_forward: graph is NULL");
  int32_t st = check_act("ngpde_gno_message_forward", act1);
  if (st) return st;
  NGPDE_REQUIRE(gno_apply_mfma_supported(cout, kdim), NGPDE_ERR_UNSUPPORTED,
                "ngpde_gno_message_forward: needs out a multiple of 16 (<= 256) and k in {16, 32, 64}, got out = %d, k = %d; compose "
                "ngpde_edge_combine_forward + ngpde_gno_apply_forward", cout, kdim);
  if (g->n_edges == 0) return NGPDE_OK;
  NGPDE_REQUIRE(t && m && (p_target || q_source || e_term), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_message_forward: NULL argument");
  return launch_gno_message_mfma_fwd(g, cout, kdim, act1, p_target, q_source, e_term, t, bh, z_out, m, (hipStream_t)stream);
}

int32_t ngpde_gat_forward(const ngpde_graph_t *g, int32_t heads, int32_t c, float negative_slope, const float *wx,
                          const float *a, float *out, float *alpha, float *al, float *ar, ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gat_forward: graph is NULL");
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(heads > 0 && c > 0 && wx && a && out && al && ar && (alpha || g->n_edges == 0), NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_gat_forward: bad arguments");
  hipStream_t stream = (hipStream_t)stream_;
  int32_t st = launch_gat_scores(g->n_nodes, heads, c, wx, a, al, ar, stream);
  if (st) return st;
  return launch_gat_fwd(g, heads, c, negative_slope, wx, al, ar, out, alpha, stream);
}

size_t ngpde_gat_workspace_bytes(const ngpde_graph_t *g, int32_t heads) {
  if (!g) return 0;
  return align256((size_t)std::max<int64_t>(g->n_edges, 1) * heads * 4) + 2 * align256((size_t)std::max<int64_t>(g->n_nodes, 1) * heads * 4) + 256;
}

int32_t ngpde_gat_backward(const ngpde_graph_t *g, int32_t heads, int32_t c, float negative_slope, const float *wx,
                           const float *a, const float *al, const float *ar, const float *alpha, const float *dout,
                           float *dwx, float *da, void *workspace, size_t workspace_bytes, ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gat_backward: graph is NULL");
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(heads > 0 && c > 0 && wx && a && al && ar && dout && dwx && da, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_gat_backward: bad arguments");
  const size_t need = ngpde_gat_workspace_bytes(g, heads);
  NGPDE_REQUIRE(workspace && workspace_bytes >= need, NGPDE_ERR_WORKSPACE,
                "ngpde_gat_backward: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
  char *ws = (char *)workspace;
  float *dscore = (float *)ws;
  float *dal = (float *)(ws + align256((size_t)std::max<int64_t>(g->n_edges, 1) * heads * 4));
  float *dar = (float *)((char *)dal + align256((size_t)g->n_nodes * heads * 4));
  return launch_gat_bwd(g, heads, c, negative_slope, wx, a, al, ar, alpha, dout, dscore, dal, dar, dwx, da,
                        (hipStream_t)stream_);
}

int32_t ngpde_gat_layer_supported(const ngpde_graph_t *g, int32_t din, int32_t heads, int32_t c) {
  return gat_layer_fused_supported(g, din, heads, c) ? 1 : 0;
}

size_t ngpde_gat_layer_workspace_bytes(const ngpde_graph_t *g, int32_t heads, int32_t c) {
  (void)c;
  return g ? gat_layer_workspace_bytes(g, heads) : 0;
}

int32_t ngpde_gat_layer_forward(const ngpde_graph_t *g, int32_t din, int32_t heads, int32_t c, float negative_slope, int32_t act,
                                const float *x, const float *weight, const float *a, const float *bias, float *y,
                                float *save_alpha, float *save_z, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gat_layer_forward: graph is NULL");
  int32_t st = check_act("ngpde_gat_layer_forward", act);
  if (st) return st;
  NGPDE_REQUIRE(gat_layer_fused_supported(g, din, heads, c), NGPDE_ERR_UNSUPPORTED,
                "ngpde_gat_layer_forward: needs din == heads * c == 64, heads in {1, 2, 4} and a graph whose tiles fit the LDS "
                "halo in both directions (got din = %d, heads = %d, c = %d); compose ngpde_dense_forward + ngpde_gat_forward", din, heads, c);
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(x && weight && a && y, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gat_layer_forward: NULL argument");
  return launch_gat_layer_fwd(g, heads, negative_slope, act, x, weight, a, bias, y, save_alpha, save_z, (hipStream_t)stream);
}

int32_t ngpde_gat_layer_backward(const ngpde_graph_t *g, int32_t din, int32_t heads, int32_t c, float negative_slope, int32_t act,
                                 const float *x, const float *weight, const float *a, const float *y_or_z, const float *save_alpha,
                                 const float *dy, float *dx, float *dweight, float *da, float *dbias, void *workspace,
                                 size_t workspace_bytes, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gat_layer_backward: graph is NULL");
  int32_t st = check_act("ngpde_gat_layer_backward", act);
  if (st) return st;
  NGPDE_REQUIRE(gat_layer_fused_supported(g, din, heads, c), NGPDE_ERR_UNSUPPORTED,
                "ngpde_gat_layer_backward: unsupported shape / graph (din = %d, heads = %d, c = %d)", din, heads, c);
  NGPDE_REQUIRE(weight && a && dweight && da && (g->n_nodes == 0 || (x && dy && (save_alpha || g->n_edges == 0))) &&
                    (act == NGPDE_ACT_IDENTITY || y_or_z || g->n_nodes == 0),
                NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gat_layer_backward: NULL argument");
  return launch_gat_layer_bwd(g, heads, negative_slope, act, x, weight, a, y_or_z, save_alpha, dy, dx, dweight, da, dbias, workspace,
                              workspace_bytes, (hipStream_t)stream);
}

int32_t ngpde_bias_act_forward(int64_t n, int32_t d, int32_t act, const float *a, const float *addend, const float *bias, float *y,
                               float *save_z, ngpde_stream_t stream) {
  NGPDE_RANGE();
  int32_t st = check_act("ngpde_bias_act_forward", act);
  if (st) return st;
  NGPDE_REQUIRE(n >= 0 && d >= 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_bias_act_forward: negative size");
  if (n == 0 || d == 0) return NGPDE_OK;
  NGPDE_REQUIRE(a && y, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_bias_act_forward: NULL argument");
  return launch_bias_act2(n, d, act, a, addend, bias, y, save_z, (hipStream_t)stream);
}

size_t ngpde_bias_act_workspace_bytes(int32_t d) { return (size_t)kColsumChunks * (size_t)std::max(d, 1) * sizeof(float); }

int32_t ngpde_bias_act_backward(int64_t n, int32_t d, int32_t act, const float *dy, const float *z, float *dz, float *dbias,
                                void *workspace, size_t workspace_bytes, ngpde_stream_t stream) {
  NGPDE_RANGE();
  int32_t st = check_act("ngpde_bias_act_backward", act);
  if (st) return st;
  NGPDE_REQUIRE(n >= 0 && d >= 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_bias_act_backward: negative size");
  if (d == 0) return NGPDE_OK;
  NGPDE_REQUIRE((n == 0 || (dy && dz)) && (act == NGPDE_ACT_IDENTITY || z || n == 0), NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_bias_act_backward: NULL argument");
  NGPDE_REQUIRE(!dbias || (workspace && workspace_bytes >= ngpde_bias_act_workspace_bytes(d)), NGPDE_ERR_WORKSPACE,
                "ngpde_bias_act_backward: workspace too small (%zu < %zu bytes)", workspace_bytes, ngpde_bias_act_workspace_bytes(d));
  if (n > 0 && !(act == NGPDE_ACT_IDENTITY && dz == dy)) {   // identity with dz aliasing dy: nothing to compute
    st = launch_dense_dz(n * d, act, dy, z ? z : dy, dz, (hipStream_t)stream);   // identity: act' = 1 whatever z is
    if (st) return st;
  }
  if (dbias) return launch_colsum2(n, d, dz, static_cast<float *>(workspace), dbias, (hipStream_t)stream);
  return NGPDE_OK;
}

int32_t ngpde_edge_mlp_supported(const ngpde_graph_t *g, int32_t h1, int32_t n_tail, const int32_t *tail_dout) {
  EdgeMlpArgs a;
  a.h1 = h1; a.n_tail = n_tail;
  if (n_tail < 0 || n_tail > 3 || (n_tail > 0 && !tail_dout)) return 0;
  int prev = h1;
  for (int l = 0; l < n_tail; ++l) { a.din[l] = prev; a.dout[l] = tail_dout[l]; prev = tail_dout[l]; }
  return edge_mlp_fused_supported(g, a) ? 1 : 0;
}

int32_t ngpde_edge_mlp_forward(const ngpde_graph_t *g, int32_t h1, int32_t act1, const float *p_target,
                               const float *q_source, const float *e_term, int32_t n_tail, const int32_t *tail_dout,
                               const int32_t *tail_act, const float *const *tail_weight, const float *const *tail_bias,
                               int32_t aggr, float *out, float *const *save_z, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_mlp_forward: graph is NULL");
  NGPDE_REQUIRE(aggr >= NGPDE_AGGR_SUM && aggr <= NGPDE_AGGR_MUL, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_edge_mlp_forward: unknown aggregation %d", aggr);
  int32_t st = check_act("ngpde_edge_mlp_forward", act1);
  if (st) return st;
  NGPDE_REQUIRE(n_tail >= 0 && n_tail <= 3, NGPDE_ERR_UNSUPPORTED, "ngpde_edge_mlp_forward: 0..3 layers after the first");
  NGPDE_REQUIRE(n_tail == 0 || (tail_dout && tail_act && tail_weight), NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_edge_mlp_forward: NULL layer table");
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(out != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_mlp_forward: out is NULL");
  EdgeMlpArgs a;
  a.h1 = h1; a.act1 = act1; a.aggr = aggr; a.n_tail = n_tail;
  a.P = p_target; a.Q = q_source; a.Eterm = e_term; a.out = out;
  int prev = h1;
  for (int l = 0; l < n_tail; ++l) {
    if ((st = check_act("ngpde_edge_mlp_forward", tail_act[l]))) return st;
    NGPDE_REQUIRE(tail_weight[l] != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_mlp_forward: layer %d weight is NULL", l);
    a.din[l] = prev; a.dout[l] = tail_dout[l]; a.act[l] = tail_act[l];
    a.wt[l] = tail_weight[l]; a.bias[l] = tail_bias ? tail_bias[l] : nullptr;
    prev = tail_dout[l];
  }
  if (save_z)
    for (int l = 0; l <= n_tail; ++l) a.save_z[l] = save_z[l];
  return launch_edge_mlp_fused_fwd(g, a, (hipStream_t)stream);
}

int32_t ngpde_edge_mlp_backward_supported(const ngpde_graph_t *g, int32_t h1, int32_t n_tail, const int32_t *tail_dout, int32_t aggr) {
  if (n_tail >= 2) return edge_mlp_deep_bwd_supported(g, h1, n_tail, tail_dout, aggr) ? 1 : 0;   // 3 / 4-layer message MLPs (edge_mlp_deep_bwd.hip)
  return edge_mlp_fused_bwd_supported(g, h1, n_tail, (n_tail == 1 && tail_dout) ? tail_dout[0] : 0, aggr) ? 1 : 0;
}

size_t ngpde_edge_mlp_backward_workspace_bytes(const ngpde_graph_t *g, int32_t h1, int32_t n_tail, const int32_t *tail_dout) {
  if (!g) return 0;
  if (n_tail >= 2) return (n_tail <= 3 && tail_dout) ? edge_mlp_deep_bwd_workspace(g, h1, n_tail, tail_dout) : 0;
  return edge_mlp_fused_bwd_workspace(g, h1, n_tail, (n_tail == 1 && tail_dout) ? tail_dout[0] : 0);
}

int32_t ngpde_edge_mlp_backward_needs_edge_buffer(const ngpde_graph_t *g, int32_t h1, int32_t act1, int32_t has_e_term, int32_t n_tail,
                                                   const int32_t *tail_dout, const int32_t *tail_act, int32_t aggr) {
  if (!g || has_e_term) return 1;
  EdgeMlpBwdArgs a;
  static const float probe = 0.f;   // (pointers are only tested for NULL here)
  a.h1 = h1; a.act1 = act1; a.aggr = aggr; a.n_tail = n_tail;
  a.P = &probe; a.Q = &probe; a.Eterm = nullptr; a.dQ = const_cast<float *>(&probe);
  if (n_tail == 1 && tail_dout && tail_act) {
    a.dw = tail_dout[0]; a.act2 = tail_act[0];
  }
  return (edge_mlp64_bwd_applicable(g, a) && edge_mlp64_bwd_dq_in_launch(g, a)) ? 0 : 1;
}

int32_t ngpde_edge_mlp_backward(const ngpde_graph_t *g, int32_t h1, int32_t act1, const float *p_target, const float *q_source,
                                const float *e_term, int32_t n_tail, const int32_t *tail_dout, const int32_t *tail_act,
                                const float *const *tail_weight, const float *const *tail_bias, int32_t aggr, const float *dout,
                                float *dp_target, float *dq_source, float *de_term, float *const *dtail_weight,
                                float *const *dtail_bias, void *workspace, size_t workspace_bytes, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_mlp_backward: graph is NULL");
  int32_t st = check_act("ngpde_edge_mlp_backward", act1);
  if (st) return st;
  NGPDE_REQUIRE(n_tail >= 0 && n_tail <= 3, NGPDE_ERR_UNSUPPORTED, "ngpde_edge_mlp_backward: at most 3 layers after the first");
  NGPDE_REQUIRE(n_tail == 0 || (tail_dout && tail_act && tail_weight && dtail_weight), NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_edge_mlp_backward: the tail layers need their width / activation / weight / gradient arrays");
  for (int l = 0; l < n_tail; ++l) {
    NGPDE_REQUIRE(tail_weight[l] && dtail_weight[l], NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_mlp_backward: tail layer %d without its weight or gradient buffer", l);
    if ((st = check_act("ngpde_edge_mlp_backward", tail_act[l]))) return st;
  }
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(dout != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_mlp_backward: dout is NULL");
  if (n_tail >= 2) {   // message MLPs of three / four layers
    EdgeMlpDeepBwdArgs d;
    d.h1 = h1; d.act1 = act1; d.aggr = aggr; d.n_tail = n_tail;
    d.P = p_target; d.Q = q_source; d.Eterm = e_term; d.dout_grad = dout;
    for (int l = 0; l < n_tail; ++l) {
      d.dout[l] = tail_dout[l]; d.act[l] = tail_act[l]; d.wt[l] = tail_weight[l]; d.bias[l] = tail_bias ? tail_bias[l] : nullptr;
      d.dwt[l] = dtail_weight[l]; d.dbias[l] = dtail_bias ? dtail_bias[l] : nullptr;
    }
    d.dP = dp_target; d.dQ = dq_source; d.dE = de_term; d.workspace = workspace; d.workspace_bytes = workspace_bytes;
    return launch_edge_mlp_deep_bwd(g, d, (hipStream_t)stream);
  }
  EdgeMlpBwdArgs a;
  a.h1 = h1; a.act1 = act1; a.aggr = aggr; a.n_tail = n_tail;
  a.P = p_target; a.Q = q_source; a.Eterm = e_term; a.dout = dout;
  if (n_tail) {
    a.dw = tail_dout[0]; a.act2 = tail_act[0]; a.wt = tail_weight[0]; a.bias = tail_bias ? tail_bias[0] : nullptr;
    a.dwt = dtail_weight[0]; a.dbias = dtail_bias ? dtail_bias[0] : nullptr;
  }
  a.dP = dp_target; a.dQ = dq_source; a.dE = de_term; a.workspace = workspace; a.workspace_bytes = workspace_bytes;
  return launch_edge_mlp_fused_bwd(g, a, (hipStream_t)stream);
}

int32_t ngpde_activation_forward(int64_t count, int32_t act, const float *z, float *a, ngpde_stream_t stream) {
  NGPDE_RANGE();
  int32_t st = check_act("ngpde_activation_forward", act);
  if (st) return st;
  if (count == 0) return NGPDE_OK;
  NGPDE_REQUIRE(z && a, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_activation_forward: NULL argument");
  return launch_activation_fwd(count, act, z, a, (hipStream_t)stream);
}

int32_t ngpde_spectral_weights(int64_t n_edges, int32_t n, const float *e, float *w, ngpde_stream_t stream) {
  NGPDE_RANGE();
  if (n_edges == 0) return NGPDE_OK;
  NGPDE_REQUIRE(e && w && n > 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_spectral_weights: bad arguments");
  return launch_spectral_weights(n_edges, (float)n, e, w, (hipStream_t)stream);
}

}  // extern "C"

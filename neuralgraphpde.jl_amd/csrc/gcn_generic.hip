// gcn_generic.hip -- any-feature-width building blocks of the GCNConv path
// (/root/reference/src/layers.jl:200-239) for shapes outside the fused kernels: CSR aggregation
// (propagate(copy_xj / w_mul_xj, g, +) and its transpose, with the layer's bias + activation tail), element-wise
// activation pullback, column sums, bias + activation.  The Dense part of that path runs on dense_mfma.hip.
// No atomics: every output element has one writer and a fixed summation order.
#include <algorithm>

#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

// ---------------------------------------------------------------------------------------------------
// generic kernels (any feature width)
// ---------------------------------------------------------------------------------------------------

// optional epilogue of the aggregation: out = act(agg + bias) with the pre-activation kept (the tail of the multiply-first
// GCNConv order, src/layers.jl:220-226)
struct SpmmTail {
  const float *bias = nullptr;
  int act = NGPDE_ACT_IDENTITY;
  float *save_z = nullptr;
};
// One wave per destination row.  Narrow rows (d <= 32) give the wave's lanes to several list entries at once -- lane =
// (entry slot, feature) -- and every lane keeps four entries in flight, so a hub row of a skewed graph (Cora-like: degree
// > 100 next to a median of 3) is a dozen load rounds instead of a hundred dependent ones; the slots' partial sums are
// combined in a fixed order (deterministic, no atomics).
template <int DP>   // features per entry slot: 64 (one slot, lanes stride over d), 32, 16, 8
__global__ __launch_bounds__(256) void spmm_generic_kernel(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                           const int *__restrict__ eid, const int2 *__restrict__ ent,
                                                           const float *__restrict__ cnorm, int self_loops, int gcn_norm,
                                                           int mean, const float *__restrict__ edge_weight, int n_nodes,
                                                           int d, const float *__restrict__ x, float *__restrict__ out,
                                                           const SpmmTail tail) {
  constexpr int SLOTS = 64 / DP;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  const int slot = lane / DP, fl = lane % DP;
  auto xat = [&](size_t idx) { return x[idx]; };
  auto load_entries = [&](int base, int &cl, float &wl) {   // ONE coalesced load per lane for the next 64 list entries
    const int pl = base + lane;
    cl = 0;
    wl = 0.f;      // past the end of the row: weight 0 on the (valid) row 0
    if (pl < re) {
      if (gcn_norm) {
        const int2 v = ent[pl];
        cl = v.x;
        wl = __int_as_float(v.y);
      } else {
        cl = col[pl];
        wl = edge_weight ? edge_weight[eid[pl]] : 1.0f;
      }
    }
  };
  constexpr int EPL = 64 / SLOTS;                 // entries per lane and 64-entry batch
  constexpr int U = EPL < 16 ? EPL : 16;          // row fetches in flight per lane
  for (int f0 = 0; f0 < d; f0 += DP) {
    const int f = f0 + fl;
    const bool fok = f < d;
    const int fc = fok ? f : 0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int cl, cn = 0;
    float wl, wn = 0.f;
    load_entries(rs, cl, wl);
    for (int base = rs; base < re; base += 64) {
      if (base + 64 < re) load_entries(base + 64, cn, wn);   // next batch in flight under this batch's row fetches
      const int cnt = min(64, re - base);
      // entries handed out by lane shuffles; all U fetches of a round are issued before the first is consumed (weight 0 and
      // row 0 for the slots past the row's end: no branch around the loads)
      for (int e0 = 0; e0 * SLOTS < cnt; e0 += U) {          // wave-uniform trip count
        float xv[U], wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = ((e0 + u) * SLOTS + slot) & 63;
          wv[u] = __shfl(wl, e);
          xv[u] = xat((size_t)__shfl(cl, e) * d + fc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u & 3] = fmaf(wv[u], xv[u], acc[u & 3]);
      }
      cl = cn;
      wl = wn;
    }
    float a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int o = DP; o < 64; o <<= 1) a += __shfl_xor(a, o);
    if (slot == 0 && fok) {
      if (gcn_norm) {
        const float ci = cnorm[row];
        if (self_loops) a = fmaf(ci, xat((size_t)row * d + f), a);
        a *= ci;
      } else if (mean) {
        const int cnt = re - rs;
        a = cnt > 0 ? a / (float)cnt : 0.f;
      }
      if (tail.bias) a += tail.bias[f];
      if (tail.save_z) tail.save_z[(size_t)row * d + f] = a;
      out[(size_t)row * d + f] = act_apply(tail.act, a);
    }
  }
}

#define NGPDE_SPMM_LAUNCH(...)                                                                                       \
  do {                                                                                                               \
    const dim3 grid_((unsigned)((g->n_nodes + 3) / 4)), block_(256);                                                 \
    if (d <= 8) hipLaunchKernelGGL(spmm_generic_kernel<8>, grid_, block_, 0, stream, __VA_ARGS__);                   \
    else if (d <= 16) hipLaunchKernelGGL(spmm_generic_kernel<16>, grid_, block_, 0, stream, __VA_ARGS__);            \
    else if (d <= 32) hipLaunchKernelGGL(spmm_generic_kernel<32>, grid_, block_, 0, stream, __VA_ARGS__);            \
    else hipLaunchKernelGGL(spmm_generic_kernel<64>, grid_, block_, 0, stream, __VA_ARGS__);                         \
  } while (0)

__global__ void act_bwd_kernel(int64_t count, int act, const float *__restrict__ dy, const float *__restrict__ z,
                               float *__restrict__ dz) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    dz[i] = dy[i] * act_deriv(act, z[i]);
}

// out[o] = sum_n a[n][o]   (bias gradient); 4 row-partials per column combined through LDS
__global__ __launch_bounds__(256) void colsum_kernel(int64_t n, int d, const float *__restrict__ a,
                                                     float *__restrict__ out) {
  __shared__ float part[4][64];
  const int o = blockIdx.x * 64 + (threadIdx.x & 63);
  const int pid = threadIdx.x >> 6;
  float s = 0.f;
  if (o < d)
    for (int64_t r = pid; r < n; r += 4) s += a[r * d + o];
  part[pid][threadIdx.x & 63] = s;
  __syncthreads();
  if (pid == 0 && o < d) out[o] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// y = act(a + addend + b), 16 bytes per thread (d % 4 == 0), optional pre-activation copy
__global__ void bias_act4_kernel(int64_t count4, int d4, int act, const float4 *__restrict__ a, const float4 *__restrict__ addend,
                                 const float4 *__restrict__ bias, float4 *__restrict__ y, float4 *__restrict__ save_z) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 z = a[i];
    if (addend) z = f4_add(z, addend[i]);
    if (bias) z = f4_add(z, bias[i % d4]);
    if (save_z) save_z[i] = z;
    y[i] = f4_act(act, z);
  }
}
__global__ void bias_act1_kernel(int64_t count, int d, int act, const float *__restrict__ a, const float *__restrict__ addend,
                                 const float *__restrict__ bias, float *__restrict__ y, float *__restrict__ save_z) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const float z = a[i] + (addend ? addend[i] : 0.f) + (bias ? bias[i % d] : 0.f);
    if (save_z) save_z[i] = z;
    y[i] = act_apply(act, z);
  }
}

// partial[chunk][o] = sum over the rows r = chunk, chunk + nchunk, ... of a[r][o]  (first stage of a column sum over many rows)
__global__ __launch_bounds__(256) void colsum_partial_kernel(int64_t n, int d, int nchunk, const float *__restrict__ a,
                                                             float *__restrict__ partial) {
  __shared__ float part[4][64];
  const int o = blockIdx.x * 64 + (threadIdx.x & 63);
  const int pid = threadIdx.x >> 6, chunk = blockIdx.y;
  float s = 0.f;
  if (o < d)
    for (int64_t r = chunk + (int64_t)pid * nchunk; r < n; r += 4 * (int64_t)nchunk) s += a[r * d + o];
  part[pid][threadIdx.x & 63] = s;
  __syncthreads();
  if (pid == 0 && o < d)
    partial[(size_t)chunk * d + o] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

}  // namespace

int32_t launch_bias_act2(int64_t n, int d, int act, const float *a, const float *addend, const float *bias, float *y, float *save_z,
                         hipStream_t stream) {
  if (n * d == 0) return NGPDE_OK;
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (d % 4 == 0 && al16(a) && al16(addend) && al16(bias) && al16(y) && al16(save_z)) {
    const int64_t c4 = n * d / 4;
    hipLaunchKernelGGL(bias_act4_kernel, dim3((unsigned)std::min<int64_t>((c4 + 255) / 256, 4096)), dim3(256), 0, stream, c4, d / 4,
                       act, reinterpret_cast<const float4 *>(a), reinterpret_cast<const float4 *>(addend),
                       reinterpret_cast<const float4 *>(bias), reinterpret_cast<float4 *>(y), reinterpret_cast<float4 *>(save_z));
  } else {
    hipLaunchKernelGGL(bias_act1_kernel, dim3((unsigned)std::min<int64_t>((n * d + 255) / 256, 4096)), dim3(256), 0, stream, n * d, d,
                       act, a, addend, bias, y, save_z);
  }
  NGPDE_LAUNCH_CHECK("bias_act kernel");
  return NGPDE_OK;
}

// column sums of a [n][d] array in two deterministic stages through `partial` ([kColsumChunks][d] floats)
int32_t launch_colsum2(int64_t n, int d, const float *a, float *partial, float *out, hipStream_t stream) {
  if (d == 0) return NGPDE_OK;
  if (n <= 4 * kColsumChunks || partial == nullptr) return launch_colsum(n, d, a, out, stream);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((d + 63) / 64, kColsumChunks), dim3(256), 0, stream, n, d, kColsumChunks, a, partial);
  NGPDE_LAUNCH_CHECK("colsum_partial_kernel");
  return launch_colsum(kColsumChunks, d, partial, out, stream);
}

int32_t launch_spmm_generic(const ngpde_graph *g, bool by_source, bool gcn_norm, int d, int aggr, const float *x,
                            const float *edge_weight, float *out, hipStream_t stream) {
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "graph is NULL");
  NGPDE_REQUIRE(!gcn_norm || g->has_norm, NGPDE_ERR_STATE, "GCN normalisation not set (call ngpde_graph_set_gcn_norm)");
  NGPDE_REQUIRE(aggr == NGPDE_AGGR_SUM || aggr == NGPDE_AGGR_MEAN, NGPDE_ERR_UNSUPPORTED,
                "aggregation %d not supported by the copy_xj path", aggr);
  if (g->n_nodes == 0 || d == 0) return NGPDE_OK;
  const Csr &c = by_source ? g->by_s : g->by_t;
  NGPDE_SPMM_LAUNCH(c.rowptr, c.col, c.eid, c.ent, g->c, g->self_loops, gcn_norm ? 1 : 0, aggr == NGPDE_AGGR_MEAN ? 1 : 0, edge_weight,
                    (int)g->n_nodes, d, x, out, SpmmTail());
  NGPDE_LAUNCH_CHECK("spmm_generic_kernel");
  return NGPDE_OK;
}

// x[i] += x[stride + i] + x[2 stride + i] + ...  (split-K partial products -> their sum, in slab order)
__global__ void sum_partials_kernel(int64_t count, int nparts, size_t stride, float *__restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    float s0 = x[i], s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 1;
    for (; z + 3 <= nparts; z += 3) {
      s1 += x[(size_t)z * stride + i];
      s2 += x[(size_t)(z + 1) * stride + i];
      s3 += x[(size_t)(z + 2) * stride + i];
    }
    for (; z < nparts; ++z) s1 += x[(size_t)z * stride + i];
    x[i] = (s0 + s1) + (s2 + s3);
  }
}
int32_t launch_sum_partials(int64_t count, int nparts, size_t stride, float *x, hipStream_t stream) {
  if (count == 0 || nparts <= 1) return NGPDE_OK;
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)std::min<int64_t>((count + 255) / 256, 4096)), dim3(256), 0, stream, count,
                     nparts, stride, x);
  NGPDE_LAUNCH_CHECK("sum_partials_kernel");
  return NGPDE_OK;
}

// GCN aggregation with the layer's tail: out = act(C (A+I) C x + bias)
int32_t launch_spmm_gcn_tail(const ngpde_graph *g, int d, const float *x, const float *bias, int act, float *out, float *save_z,
                             hipStream_t stream) {
  NGPDE_REQUIRE(g && g->has_norm, NGPDE_ERR_STATE, "GCN normalisation not set (call ngpde_graph_set_gcn_norm)");
  if (g->n_nodes == 0 || d == 0) return NGPDE_OK;
  const Csr &c = g->by_t;
  SpmmTail t;
  t.bias = bias; t.act = act; t.save_z = save_z;
  NGPDE_SPMM_LAUNCH(c.rowptr, c.col, c.eid, c.ent, g->c, g->self_loops, 1, 0, (const float *)nullptr, (int)g->n_nodes, d, x, out, t);
  NGPDE_LAUNCH_CHECK("spmm_generic_kernel (tail)");
  return NGPDE_OK;
}

int32_t launch_act_bwd(int64_t count, int act, const float *dy, const float *z, float *dz, hipStream_t stream) {
  if (count == 0) return NGPDE_OK;
  const int blocks = (int)std::min<int64_t>((count + 255) / 256, 2048);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, stream, count, act, dy, z, dz);
  NGPDE_LAUNCH_CHECK("act_bwd_kernel");
  return NGPDE_OK;
}

int32_t launch_colsum(int64_t n, int d, const float *a, float *out, hipStream_t stream) {
  if (d == 0) return NGPDE_OK;
  hipLaunchKernelGGL(colsum_kernel, dim3((d + 63) / 64), dim3(256), 0, stream, n, d, a, out);
  NGPDE_LAUNCH_CHECK("colsum_kernel");
  return NGPDE_OK;
}

// ---- gradient w.r.t. the edge_weight ARGUMENT of GCNConv (/root/reference/src/layers.jl:206-231) ------------------------------
// With xp = the array entering the propagation (x, or x W when Dout < Din), x3 = c_i sum_e w_e c_s xp_s its output and g3 = dL/dx3:
//   dL/dw_e = c_t c_s (g3_t . xp_s)  [through e_mul_xj, :228]  -  1/2 c_t^2 (g3_t . x3_t + dxp_t . xp_t)  [through the weighted
//   degree d_t = sum w_e, c = d^(-1/2), :224-226 and :234], dxp = dL/dxp (the layer's own input gradient).
// node_term: nd[i] = -1/2 c_i^2 (g3_i . x3_i + dxp_i . xp_i), one 16-lane group per node; x3 may be given as z - bias.
__global__ void gcn_ew_node_term_kernel(int n, int d, const float *__restrict__ c, const float *__restrict__ g3, const float *__restrict__ x3,
                                        const float *__restrict__ bias, const float *__restrict__ dxp, const float *__restrict__ xp,
                                        float *__restrict__ nd) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, q = threadIdx.x & 15;
  if (i >= n) return;   // (whole 16-lane groups leave together)
  float s = 0.f;
  for (int f = q; f < d; f += 16) {
    const size_t k = (size_t)i * d + f;
    s += g3[k] * (x3[k] - (bias ? bias[f] : 0.f)) + dxp[k] * xp[k];
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 16);
  if (q == 0) nd[i] = -0.5f * c[i] * c[i] * s;
}
// one wave per target row: its four 16-lane groups take the row's entries in turn; dw[eid] = c_t c_s (g3_t . xp_s) + nd[t]
__global__ void gcn_ew_edge_kernel(int n, int d, const int *__restrict__ rowptr, const int *__restrict__ col, const int *__restrict__ eid,
                                   const float *__restrict__ c, const float *__restrict__ g3, const float *__restrict__ xp,
                                   const float *__restrict__ nd, float *__restrict__ dw) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n) return;
  const int lane = threadIdx.x & 63, grp = lane >> 4, q = lane & 15;
  const int p0 = rowptr[i], p1 = rowptr[i + 1];
  const float ci = c[i], ndi = nd[i];
  for (int p = p0 + grp; p < p1; p += 4) {
    const int j = col[p];
    float s = 0.f;
    for (int f = q; f < d; f += 16) s += g3[(size_t)i * d + f] * xp[(size_t)j * d + f];
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 16);
    if (q == 0) dw[eid[p]] = ci * c[j] * s + ndi;
  }
}

int32_t launch_gcn_edge_weight_grad(const ngpde_graph *g, int d, const float *g3, const float *x3, const float *bias_or_null, const float *dxp,
                                    const float *xp, float *nd_scratch, float *dw, hipStream_t stream) {
  NGPDE_REQUIRE(g && g->has_norm, NGPDE_ERR_STATE, "GCN normalisation not set (call ngpde_graph_set_gcn_norm)");
  const int n = (int)g->n_nodes;
  if (n == 0 || g->n_edges == 0) return NGPDE_OK;
  hipLaunchKernelGGL(gcn_ew_node_term_kernel, dim3((unsigned)(((int64_t)n * 16 + 255) / 256)), dim3(256), 0, stream, n, d, g->c, g3, x3,
                     bias_or_null, dxp, xp, nd_scratch);
  NGPDE_LAUNCH_CHECK("gcn_ew_node_term_kernel");
  hipLaunchKernelGGL(gcn_ew_edge_kernel, dim3((unsigned)(((int64_t)n * 64 + 255) / 256)), dim3(256), 0, stream, n, d, g->by_t.rowptr,
                     g->by_t.col, g->by_t.eid, g->c, g3, xp, nd_scratch, dw);
  NGPDE_LAUNCH_CHECK("gcn_ew_edge_kernel");
  return NGPDE_OK;
}

}  // namespace ngpde

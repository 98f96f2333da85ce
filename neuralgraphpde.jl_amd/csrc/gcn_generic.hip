// gcn_generic.hip -- any-feature-width building blocks of the GCNConv path
// (/root/reference/src/layers.jl:200-239) for shapes outside the fused kernels: CSR aggregation
// (propagate(copy_xj / w_mul_xj, g, +) and its transpose), dense forward / backward, bias + activation.
// No atomics: every output element has one writer and a fixed summation order.
#include <algorithm>

#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

// ---------------------------------------------------------------------------------------------------
// generic kernels (any feature width)
// ---------------------------------------------------------------------------------------------------

// one wave per destination row; lanes stride over the features; CSR order summation, no atomics
__global__ __launch_bounds__(256) void spmm_generic_kernel(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                           const int *__restrict__ eid, const int2 *__restrict__ ent,
                                                           const float *__restrict__ cnorm, int self_loops, int gcn_norm,
                                                           int mean, const float *__restrict__ edge_weight, int n_nodes,
                                                           int d, const float *__restrict__ x, float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  for (int f = lane; f < d; f += 64) {
    float acc = 0.f;
    for (int p = rs; p < re; ++p) {
      float w;
      int c;
      if (gcn_norm) {
        const int2 v = ent[p];
        c = v.x;
        w = __int_as_float(v.y);
      } else {
        c = col[p];
        w = edge_weight ? edge_weight[eid[p]] : 1.0f;
      }
      acc = fmaf(w, x[(size_t)c * d + f], acc);
    }
    if (gcn_norm) {
      const float ci = cnorm[row];
      if (self_loops) acc = fmaf(ci, x[(size_t)row * d + f], acc);
      acc *= ci;
    } else if (mean) {
      const int cnt = re - rs;
      acc = cnt > 0 ? acc / (float)cnt : 0.f;
    }
    out[(size_t)row * d + f] = acc;
  }
}

// y[n][o] = act(sum_i x[n][i] wt[i][o] + b[o]);  16x16 output tile per workgroup, K staged through LDS
__global__ __launch_bounds__(256) void dense_fwd_kernel(int64_t n, int din, int dout, int act,
                                                        const float *__restrict__ x, const float *__restrict__ wt,
                                                        const float *__restrict__ bias, float *__restrict__ y,
                                                        float *__restrict__ save_z) {
  __shared__ float xs[16][17], ws[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t row = (int64_t)blockIdx.x * 16 + ty;
  const int o = blockIdx.y * 16 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < din; k0 += 16) {
    xs[ty][tx] = (row < n && k0 + tx < din) ? x[row * din + k0 + tx] : 0.f;
    ws[ty][tx] = (k0 + ty < din && o < dout) ? wt[(size_t)(k0 + ty) * dout + o] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fmaf(xs[ty][k], ws[k][tx], acc);
    __syncthreads();
  }
  if (row < n && o < dout) {
    const float z = acc + (bias ? bias[o] : 0.f);
    if (save_z) save_z[row * dout + o] = z;
    y[row * dout + o] = act_apply(act, z);
  }
}

__global__ void act_bwd_kernel(int64_t count, int act, const float *__restrict__ dy, const float *__restrict__ z,
                               float *__restrict__ dz) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    dz[i] = dy[i] * act_deriv(act, z[i]);
}

// dx[n][i] = sum_o dz[n][o] wt[i][o]
__global__ __launch_bounds__(256) void dense_bwd_input_kernel(int64_t n, int din, int dout,
                                                              const float *__restrict__ dz,
                                                              const float *__restrict__ wt, float *__restrict__ dx) {
  __shared__ float zs[16][17], ws[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t row = (int64_t)blockIdx.x * 16 + ty;
  const int i = blockIdx.y * 16 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < dout; k0 += 16) {
    zs[ty][tx] = (row < n && k0 + tx < dout) ? dz[row * dout + k0 + tx] : 0.f;
    // ws[k][j] = wt[i0 + j][k0 + k]
    const int wi = blockIdx.y * 16 + ty;
    ws[tx][ty] = (wi < din && k0 + tx < dout) ? wt[(size_t)wi * dout + k0 + tx] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fmaf(zs[ty][k], ws[k][tx], acc);
    __syncthreads();
  }
  if (row < n && i < din) dx[row * din + i] = acc;
}

// dwt[i][o] = sum_n x[n][i] dz[n][o]; one workgroup per 16x16 tile of dwt, loops over all rows
// (deterministic: fixed summation order)
__global__ __launch_bounds__(256) void dense_bwd_weight_kernel(int64_t n, int din, int dout,
                                                               const float *__restrict__ x,
                                                               const float *__restrict__ dz, float *__restrict__ dwt) {
  __shared__ float xs[16][17], zs[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + ty;   // row of dwt
  const int o = blockIdx.y * 16 + tx;   // col of dwt
  float acc = 0.f;
  for (int64_t n0 = 0; n0 < n; n0 += 16) {
    // xs[k][j] = x[n0 + k][i0 + j];  zs[k][j] = dz[n0 + k][o0 + j]
    xs[ty][tx] = (n0 + ty < n && blockIdx.x * 16 + tx < din) ? x[(n0 + ty) * din + blockIdx.x * 16 + tx] : 0.f;
    zs[ty][tx] = (n0 + ty < n && o < dout) ? dz[(n0 + ty) * dout + o] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fmaf(xs[k][ty], zs[k][tx], acc);
    __syncthreads();
  }
  if (i < din && o < dout) dwt[(size_t)i * dout + o] = acc;
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

// y = act(a + b) row-wise bias, optional pre-activation copy (generic dout < din path)
__global__ void bias_act_kernel(int64_t n, int d, int act, const float *__restrict__ a, const float *__restrict__ bias,
                                float *__restrict__ y, float *__restrict__ save_z) {
  const int64_t count = n * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const float z = a[i] + (bias ? bias[i % d] : 0.f);
    if (save_z) save_z[i] = z;
    y[i] = act_apply(act, z);
  }
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
  hipLaunchKernelGGL(spmm_generic_kernel, dim3((unsigned)((g->n_nodes + 3) / 4)), dim3(256), 0, stream, c.rowptr,
                     c.col, c.eid, c.ent, g->c, g->self_loops, gcn_norm ? 1 : 0, aggr == NGPDE_AGGR_MEAN ? 1 : 0,
                     edge_weight, (int)g->n_nodes, d, x, out);
  NGPDE_LAUNCH_CHECK("spmm_generic_kernel");
  return NGPDE_OK;
}

int32_t launch_dense_fwd(int64_t n, int din, int dout, int act, const float *x, const float *wt, const float *bias,
                         float *y, float *save_z, hipStream_t stream) {
  if (n == 0 || dout == 0) return NGPDE_OK;
  hipLaunchKernelGGL(dense_fwd_kernel, dim3((unsigned)((n + 15) / 16), (dout + 15) / 16), dim3(256), 0, stream, n, din,
                     dout, act, x, wt, bias, y, save_z);
  NGPDE_LAUNCH_CHECK("dense_fwd_kernel");
  return NGPDE_OK;
}

int32_t launch_act_bwd(int64_t count, int act, const float *dy, const float *z, float *dz, hipStream_t stream) {
  if (count == 0) return NGPDE_OK;
  const int blocks = (int)std::min<int64_t>((count + 255) / 256, 2048);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, stream, count, act, dy, z, dz);
  NGPDE_LAUNCH_CHECK("act_bwd_kernel");
  return NGPDE_OK;
}

int32_t launch_dense_bwd_input(int64_t n, int din, int dout, const float *dz, const float *wt, float *dx,
                               hipStream_t stream) {
  if (n == 0 || din == 0) return NGPDE_OK;
  hipLaunchKernelGGL(dense_bwd_input_kernel, dim3((unsigned)((n + 15) / 16), (din + 15) / 16), dim3(256), 0, stream, n,
                     din, dout, dz, wt, dx);
  NGPDE_LAUNCH_CHECK("dense_bwd_input_kernel");
  return NGPDE_OK;
}

int32_t launch_dense_bwd_weight(int64_t n, int din, int dout, const float *x, const float *dz, float *dwt,
                                hipStream_t stream) {
  if (din == 0 || dout == 0) return NGPDE_OK;
  hipLaunchKernelGGL(dense_bwd_weight_kernel, dim3((din + 15) / 16, (dout + 15) / 16), dim3(256), 0, stream, n, din,
                     dout, x, dz, dwt);
  NGPDE_LAUNCH_CHECK("dense_bwd_weight_kernel");
  return NGPDE_OK;
}

int32_t launch_colsum(int64_t n, int d, const float *a, float *out, hipStream_t stream) {
  if (d == 0) return NGPDE_OK;
  hipLaunchKernelGGL(colsum_kernel, dim3((d + 63) / 64), dim3(256), 0, stream, n, d, a, out);
  NGPDE_LAUNCH_CHECK("colsum_kernel");
  return NGPDE_OK;
}

int32_t launch_bias_act(int64_t n, int d, int act, const float *a, const float *bias, float *y, float *save_z,
                        hipStream_t stream) {
  if (n * d == 0) return NGPDE_OK;
  const int blocks = (int)std::min<int64_t>((n * d + 255) / 256, 2048);
  hipLaunchKernelGGL(bias_act_kernel, dim3(blocks), dim3(256), 0, stream, n, d, act, a, bias, y, save_z);
  NGPDE_LAUNCH_CHECK("bias_act_kernel");
  return NGPDE_OK;
}

}  // namespace ngpde

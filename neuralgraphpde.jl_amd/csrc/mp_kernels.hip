// mp_kernels.hip -- message-passing primitives behind the edge-MLP layers of /root/reference/src/layers.jl
// (ExplicitEdgeConv :94-112, VMHConv :308-332, MPPDEConv :390-422, GNOConv :509-547) and the GAT-style
// softmax aggregation (primitive re-exported at src/NeuralGraphPDE.jl:7).
//
// `propagate(message, g, aggr; xi, xj, e)` [GraphNeuralNetworks.jl] = gather at t / gather at s -> message on
// the whole edge set -> scatter(aggr) at t.  Here per-edge arrays live in CSR-by-target order ("p order":
// the edges of one target are contiguous, in COO order), so
//   * the gathers + the first Dense layer of the message MLP collapse into  z_p = P[t_p] + Q[s_p] + E_p
//     with node-level P, Q (the first layer is linear in the concatenated blocks:
//     W [hi; hj; di - dj; e; theta] = (Wa hi + Wc di + We theta + b) + (Wb hj - Wc dj) + Wd e),
//   * scatter(aggr) is an atomic-free segmented reduction over contiguous rows,
//   * pullbacks towards source nodes walk the CSR-by-source list through `xpos` (the p position of each entry).
// First correct versions: one wave per row, lanes stride the feature axis; no atomics, fixed summation
// order (bitwise reproducible).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

// ---- concat-free dense:  y = act([X1 | X2 | ...] Wt + b) ------------------------------------------------
// The reference builds vcat(...) temporaries ((sum D) x E, 1.66 GB per GPU shard at C4); here the blocks are
// read in place.  A block with row_div > 1 is a per-graph feature: row r reads row r / row_div
// (repeat(theta; inner=(1, E / G)), src/layers.jl:410,:418).

// dz = dy * act'(z)
__global__ void dense_dz_kernel(int64_t count, int act, const float *__restrict__ dy, const float *__restrict__ z,
                                float *__restrict__ dz) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    dz[i] = dy[i] * act_deriv(act, z[i]);
}

// ---- edge-order helpers ------------------------------------------------------------------------------------

// dst[p] = src[eid[p]]  (COO order -> p order), or the inverse scatter when `inverse`
__global__ void edge_permute_kernel(int64_t n_edges, int d, const int *__restrict__ eid, int inverse,
                                    const float *__restrict__ src, float *__restrict__ dst) {
  const int64_t total = n_edges * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / d;
    const int f = (int)(i % d);
    const int64_t e = eid[p];
    if (inverse) dst[e * d + f] = src[p * d + f];
    else dst[p * d + f] = src[e * d + f];
  }
}

// z_p = P[t] + Q[s_p] (+ E_p);  a_p = act(z_p).  One wave per target row.
__global__ __launch_bounds__(256) void edge_combine_fwd_kernel(int n_nodes, int h, int act, const int *__restrict__ rowptr,
                                                               const int *__restrict__ col, const float *__restrict__ P,
                                                               const float *__restrict__ Q, const float *__restrict__ Eterm,
                                                               float *__restrict__ a_out, float *__restrict__ z_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  for (int f = lane; f < h; f += 64) {
    const float pv = P ? P[(size_t)row * h + f] : 0.f;
    for (int p = rs; p < re; ++p) {
      float z = pv + (Q ? Q[(size_t)col[p] * h + f] : 0.f);
      if (Eterm) z += Eterm[(size_t)p * h + f];
      if (z_out) z_out[(size_t)p * h + f] = z;
      a_out[(size_t)p * h + f] = act_apply(act, z);
    }
  }
}

// ---- 16-byte forms of the per-row edge kernels (feature width a multiple of 4) -------------------------------------------
// One wave per target row; the wave's lanes are (entry slot, float4 column): DPL lanes cover one entry's row, 64 / DPL
// entries are processed at once and every lane keeps four entries in flight.  A row's entries are contiguous in p order, so
// the per-edge arrays stream; partial sums of the slots are combined in a fixed order.
template <int DPL>
__global__ __launch_bounds__(256) void edge_combine_fwd4_kernel(int n_nodes, int c4n, int act, const int *__restrict__ rowptr,
                                                                const int *__restrict__ col, const float4 *__restrict__ P,
                                                                const float4 *__restrict__ Q, const float4 *__restrict__ Eterm,
                                                                float4 *__restrict__ a_out, float4 *__restrict__ z_out) {
  constexpr int SLOTS = 64 / DPL;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  const int slot = lane / DPL, c4 = lane % DPL;
  if (c4 >= c4n) return;
  const float4 pv = P ? P[(size_t)row * c4n + c4] : f4_zero();
  for (int p0 = rs + slot; p0 < re; p0 += 4 * SLOTS) {
    float4 z[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = min(p0 + u * SLOTS, re - 1);
      z[u] = pv;
      if (Q) z[u] = f4_add(z[u], Q[(size_t)col[p] * c4n + c4]);
      if (Eterm) z[u] = f4_add(z[u], Eterm[(size_t)p * c4n + c4]);
    }
    float4 a[4] = {z[0], z[1], z[2], z[3]};
    f4n_act<4>(act, a);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * SLOTS;
      if (p < re) {
        if (z_out) z_out[(size_t)p * c4n + c4] = z[u];
        a_out[(size_t)p * c4n + c4] = a[u];
      }
    }
  }
}

// sum / mean of the row's entries
template <int DPL>
__global__ __launch_bounds__(256) void segment_sum4_kernel(int n_nodes, int c4n, int mean, const int *__restrict__ rowptr,
                                                           const float4 *__restrict__ M, float4 *__restrict__ out) {
  constexpr int SLOTS = 64 / DPL;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  const int slot = lane / DPL, c4 = min(lane % DPL, c4n - 1);
  float4 acc[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
  for (int p0 = rs + slot; p0 < re; p0 += 4 * SLOTS) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * SLOTS;
      v[u] = p < re ? M[(size_t)p * c4n + c4] : f4_zero();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f4_add(acc[u], v[u]);
  }
  float4 a = f4_add(f4_add(acc[0], acc[1]), f4_add(acc[2], acc[3]));
#pragma unroll
  for (int o = DPL; o < 64; o <<= 1)
    a = f4_add(a, make_float4(__shfl_xor(a.x, o), __shfl_xor(a.y, o), __shfl_xor(a.z, o), __shfl_xor(a.w, o)));
  if (mean) a = re > rs ? f4_scale(1.0f / (float)(re - rs), a) : f4_zero();
  if (slot == 0 && lane % DPL < c4n) out[(size_t)row * c4n + c4] = a;
}

// dz_p = da_p * act'(z_p);  dP[t] = sum over the row
template <int DPL>
__global__ __launch_bounds__(256) void edge_combine_bwd_target4_kernel(int n_nodes, int c4n, int act, const int *__restrict__ rowptr,
                                                                       const float4 *__restrict__ da, const float4 *__restrict__ z,
                                                                       float4 *__restrict__ dz, float4 *__restrict__ dP) {
  constexpr int SLOTS = 64 / DPL;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  const int slot = lane / DPL, c4 = min(lane % DPL, c4n - 1);
  const bool cok = lane % DPL < c4n;
  float4 acc = f4_zero();
  for (int p0 = rs + slot; p0 < re; p0 += 4 * SLOTS) {
    float4 g[4], zz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = min(p0 + u * SLOTS, re - 1);
      g[u] = da[(size_t)p * c4n + c4];
      zz[u] = z ? z[(size_t)p * c4n + c4] : f4_zero();
    }
    if (z) {
      f4n_dact<4>(act, zz);
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] = f4_mul(g[u], zz[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * SLOTS;
      if (p < re) {
        if (cok) dz[(size_t)p * c4n + c4] = g[u];
        acc = f4_add(acc, g[u]);
      }
    }
  }
#pragma unroll
  for (int o = DPL; o < 64; o <<= 1)
    acc = f4_add(acc, make_float4(__shfl_xor(acc.x, o), __shfl_xor(acc.y, o), __shfl_xor(acc.z, o), __shfl_xor(acc.w, o)));
  if (dP && slot == 0 && cok) dP[(size_t)row * c4n + c4] = acc;
}

// lanes per entry for a row of c4n float4: the next power of two (4 .. 64)
inline int dpl_for(int c4n) { return c4n <= 4 ? 4 : c4n <= 8 ? 8 : c4n <= 16 ? 16 : c4n <= 32 ? 32 : 64; }
inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
#define NGPDE_DPL_DISPATCH(KERNEL, c4n, grid, stream, ...)                                               \
  switch (dpl_for(c4n)) {                                                                                \
    case 4: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(256), 0, stream, __VA_ARGS__); break;                \
    case 8: hipLaunchKernelGGL(KERNEL<8>, grid, dim3(256), 0, stream, __VA_ARGS__); break;                \
    case 16: hipLaunchKernelGGL(KERNEL<16>, grid, dim3(256), 0, stream, __VA_ARGS__); break;              \
    case 32: hipLaunchKernelGGL(KERNEL<32>, grid, dim3(256), 0, stream, __VA_ARGS__); break;              \
    default: hipLaunchKernelGGL(KERNEL<64>, grid, dim3(256), 0, stream, __VA_ARGS__); break;              \
  }

// dz_p = da_p * act'(z_p) (in place into dz);  dP[t] = sum over the row
__global__ __launch_bounds__(256) void edge_combine_bwd_target_kernel(int n_nodes, int h, int act, const int *__restrict__ rowptr,
                                                                      const float *__restrict__ da, const float *__restrict__ z,
                                                                      float *__restrict__ dz, float *__restrict__ dP) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  for (int f = lane; f < h; f += 64) {
    float s = 0.f;
    for (int p = rs; p < re; ++p) {
      const float g = da[(size_t)p * h + f] * (z ? act_deriv(act, z[(size_t)p * h + f]) : 1.0f);
      dz[(size_t)p * h + f] = g;
      s += g;
    }
    if (dP) dP[(size_t)row * h + f] = s;
  }
}

// dQ[s] = sum over the edges leaving s of dz_p  (CSR by source, xpos = p position of each entry)
__global__ __launch_bounds__(256) void edge_sum_by_source_kernel(int n_nodes, int h, const int *__restrict__ rowptr_s,
                                                                 const int *__restrict__ xpos, const float *__restrict__ dz,
                                                                 float *__restrict__ dQ) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr_s[row], re = rowptr_s[row + 1];
  for (int f = lane; f < h; f += 64) {
    float s = 0.f;
    for (int q0 = rs; q0 < re; q0 += 8) {   // 8 independent row reads in flight, summed in list order
      int pp[8];
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pp[u] = (q0 + u < re) ? xpos[q0 + u] : -1;
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (pp[u] >= 0) ? dz[(size_t)pp[u] * h + f] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    dQ[(size_t)row * h + f] = s;
  }
}

// ---- aggregate_neighbors: out[i] = aggr_{p in row i} M[p]   (scatter(aggr, m, t) of NNlib) --------------
__global__ __launch_bounds__(256) void segment_reduce_fwd_kernel(int n_nodes, int d, int aggr, const int *__restrict__ rowptr,
                                                                 const float *__restrict__ M, float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  for (int f = lane; f < d; f += 64) {
    float acc;
    if (aggr == NGPDE_AGGR_MAX) acc = -INFINITY;
    else if (aggr == NGPDE_AGGR_MIN) acc = INFINITY;
    else if (aggr == NGPDE_AGGR_MUL) acc = 1.f;     // scatter(*) starts from the neutral element: an empty neighbourhood gives 1
    else acc = 0.f;
    for (int p = rs; p < re; ++p) {
      const float v = M[(size_t)p * d + f];
      if (aggr == NGPDE_AGGR_MAX) acc = fmaxf(acc, v);
      else if (aggr == NGPDE_AGGR_MIN) acc = fminf(acc, v);
      else if (aggr == NGPDE_AGGR_MUL) acc *= v;
      else acc += v;
    }
    if (aggr == NGPDE_AGGR_MEAN) acc = (re > rs) ? acc / (float)(re - rs) : 0.f;   // mean of an empty neighbourhood is 0
    out[(size_t)row * d + f] = acc;
  }
}

__global__ __launch_bounds__(256) void segment_reduce_bwd_kernel(int n_nodes, int d, int aggr, const int *__restrict__ rowptr,
                                                                 const float *__restrict__ M, const float *__restrict__ out,
                                                                 const float *__restrict__ dout, float *__restrict__ dM) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  for (int f = lane; f < d; f += 64) {
    const float g = dout[(size_t)row * d + f];
    const float scale = (aggr == NGPDE_AGGR_MEAN && re > rs) ? 1.0f / (float)(re - rs) : 1.0f;
    const float ext = (aggr == NGPDE_AGGR_MAX || aggr == NGPDE_AGGR_MIN) ? out[(size_t)row * d + f] : 0.f;
    for (int p = rs; p < re; ++p) {
      float v = g * scale;
      if (aggr == NGPDE_AGGR_MAX || aggr == NGPDE_AGGR_MIN) v = (M[(size_t)p * d + f] == ext) ? g : 0.f;  // NNlib: every extremal entry
      if (aggr == NGPDE_AGGR_MUL) {   // d(prod)/dM[p] = product of the row's other entries (NNlib: dout .* out ./ M, without the division)
        float others = 1.f;
        for (int p2 = rs; p2 < re; ++p2) others *= (p2 == p) ? 1.f : M[(size_t)p2 * d + f];
        v = g * others;
      }
      dM[(size_t)p * d + f] = v;
    }
  }
}

// ---- GNOConv contraction: m_p[o] = sum_i K_p[o + out*i] h[s_p][i]   (batched_mul, src/layers.jl:527-530) -----
__global__ __launch_bounds__(256) void gno_contract_fwd_kernel(int64_t n_edges, int cin, int cout, const int *__restrict__ col,
                                                               const float *__restrict__ K, const float *__restrict__ hfeat,
                                                               float *__restrict__ m) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_edges) return;
  const float *Kp = K + (size_t)p * cin * cout;
  const float *hp = hfeat + (size_t)col[p] * cin;
  for (int o = lane; o < cout; o += 64) {
    float acc = 0.f;
    for (int i = 0; i < cin; ++i) acc = fmaf(Kp[o + (size_t)cout * i], hp[i], acc);   // coalesced over o
    m[(size_t)p * cout + o] = acc;
  }
}

// dK_p[o + out*i] = dm_p[o] h[s_p][i];  dhe_p[i] = sum_o K_p[o + out*i] dm_p[o]  (per-edge; summed by source afterwards)
__global__ __launch_bounds__(256) void gno_contract_bwd_kernel(int64_t n_edges, int cin, int cout, const int *__restrict__ col,
                                                               const float *__restrict__ K, const float *__restrict__ hfeat,
                                                               const float *__restrict__ dm, float *__restrict__ dK,
                                                               float *__restrict__ dhe) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_edges) return;
  const float *Kp = K + (size_t)p * cin * cout;
  const float *hp = hfeat + (size_t)col[p] * cin;
  const float *dmp = dm + (size_t)p * cout;
  for (int i = 0; i < cin; ++i) {
    const float hv = hp[i];
    float part = 0.f;
    for (int o = lane; o < cout; o += 64) {
      const float g = dmp[o];
      if (dK) dK[(size_t)p * cin * cout + o + (size_t)cout * i] = g * hv;
      part = fmaf(Kp[o + (size_t)cout * i], g, part);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0 && dhe) dhe[(size_t)p * cin + i] = part;
  }
}

// ---- GNOConv, reassociated: with z_e the last hidden activation of phi (width k) and W2, b2 its
// last Dense layer, K_e = reshape(W2 z_e + b2, out, in) and  m_e = K_e h_j = T_j z_e + (B2 h_j),
// T_j[o][kk] = sum_i W2[o + out*i][kk] h_j[i]  a NODE-level product.  The in*out x E kernel tensor (64 KB per edge at
// 128-d) is never formed.  One workgroup per SOURCE node stages T_j in LDS (row stride out + 1: conflict-free for
// both access directions) and serves all edges leaving j; outputs land at the edges' p positions.
constexpr int kGnoBatch = 16;   // edges of one source staged per LDS pass

__device__ __forceinline__ int pad4(int v) { return (v + 3) & ~3; }

// stage T_j ([cout][kdim] row-major in memory) as Tl[kk][o] with zero padding to multiples of 4 in both extents
__device__ __forceinline__ void gno_stage_t(float *Tl, const float *__restrict__ Tj, int cout, int kdim, int cP, int kdP, int tid) {
  const int ts = cP + 1;
  for (int idx = tid; idx < cP * kdP; idx += 256) {
    const int o = idx / kdP, kk = idx - o * kdP;
    Tl[kk * ts + o] = (o < cout && kk < kdim) ? Tj[(size_t)o * kdim + kk] : 0.f;
  }
}

// stage up to kGnoBatch edge rows (width w, padded to wP with zeros) addressed through xpos
__device__ __forceinline__ void gno_stage_rows(float *dst, const float *__restrict__ src, const int *__restrict__ xpos, int q0, int nb,
                                               int w, int wP, int tid) {
  for (int idx = tid; idx < nb * wP; idx += 256) {
    const int e2 = idx / wP, c = idx - e2 * wP;
    dst[idx] = c < w ? src[(size_t)xpos[q0 + e2] * w + c] : 0.f;
  }
}

__global__ __launch_bounds__(256) void gno_apply_fwd_kernel(int n_nodes, int cout, int kdim, const int *__restrict__ rowptr_s,
                                                            const int *__restrict__ xpos, const float *__restrict__ T,
                                                            const float *__restrict__ Bh, const float *__restrict__ z,
                                                            float *__restrict__ m) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int cP = pad4(cout), kdP = pad4(kdim), ts = cP + 1;
  float *zl = sh;                                   // [kGnoBatch][kdP]   (16-byte aligned rows)
  float *Tl = sh + kGnoBatch * kdP;                 // [kdP][cP + 1]
  __shared__ int pl[kGnoBatch];
  const int j = blockIdx.x, tid = threadIdx.x;
  const int rs = rowptr_s[j], re = rowptr_s[j + 1];
  if (rs == re) return;
  gno_stage_t(Tl, T + (size_t)j * cout * kdim, cout, kdim, cP, kdP, tid);
  const int EB = 256 / cout;                        // edges served concurrently (cout <= 256)
  const int el = tid / cout, o = tid - el * cout;
  const float bias = (Bh && el < EB) ? Bh[(size_t)j * cout + o] : 0.f;
  for (int q0 = rs; q0 < re; q0 += kGnoBatch) {
    const int nb = min(kGnoBatch, re - q0);
    __syncthreads();
    if (tid < nb) pl[tid] = xpos[q0 + tid];
    gno_stage_rows(zl, z, xpos, q0, nb, kdim, kdP, tid);
    __syncthreads();
    if (el < EB) {
      for (int e0 = el; e0 < nb; e0 += 4 * EB) {    // register tile: 4 edges share every T read
        int er[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) er[i] = min(e0 + i * EB, nb - 1);
        float acc[4] = {bias, bias, bias, bias};
        for (int kk = 0; kk < kdP; kk += 4) {
          const float t0 = Tl[kk * ts + o], t1 = Tl[(kk + 1) * ts + o], t2 = Tl[(kk + 2) * ts + o], t3 = Tl[(kk + 3) * ts + o];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 z4 = *reinterpret_cast<const float4 *>(&zl[er[i] * kdP + kk]);
            acc[i] = fmaf(t0, z4.x, fmaf(t1, z4.y, fmaf(t2, z4.z, fmaf(t3, z4.w, acc[i]))));
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (e0 + i * EB < nb) m[(size_t)pl[e0 + i * EB] * cout + o] = acc[i];
      }
    }
  }
}

// pullback: dz_e = T_j^T dm_e;  dT_j = sum_{e leaving j} dm_e (x) z_e;  dBh_j = sum_e dm_e.
// Thread (kk = tid mod KP, og = tid / KP), KP = kdim rounded up to a power of two, keeps the A consecutive rows
// dT_j[og*A .. og*A + A)[kk] in registers (A <= 32, a multiple of 4).
__global__ __launch_bounds__(256) void gno_apply_bwd_kernel(int n_nodes, int cout, int kdim, int kp_log2, int A,
                                                            const int *__restrict__ rowptr_s, const int *__restrict__ xpos,
                                                            const float *__restrict__ T, const float *__restrict__ z,
                                                            const float *__restrict__ dm, float *__restrict__ dT,
                                                            float *__restrict__ dBh, float *__restrict__ dz) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int cP = pad4(cout), kdP = pad4(kdim), ts = cP + 1;
  float *zl = sh;                                   // [kGnoBatch][kdP]
  float *dml = zl + kGnoBatch * kdP;                // [kGnoBatch][cP]
  float *Tl = dml + kGnoBatch * cP;                 // [kdP][cP + 1]
  __shared__ int pl[kGnoBatch];
  const int j = blockIdx.x, tid = threadIdx.x;
  const int rs = rowptr_s[j], re = rowptr_s[j + 1];
  const int total = cout * kdim;
  const int kk = tid & ((1 << kp_log2) - 1), og = tid >> kp_log2, OG = 256 >> kp_log2;
  constexpr int MAXA = 32;
  float acc[MAXA];
#pragma unroll
  for (int a = 0; a < MAXA; ++a) acc[a] = 0.f;
  float accb = 0.f;
  if (dz && rs < re) gno_stage_t(Tl, T + (size_t)j * total, cout, kdim, cP, kdP, tid);
  const int ob = og * A;                            // first output row of this thread's block
  for (int q0 = rs; q0 < re; q0 += kGnoBatch) {
    const int nb = min(kGnoBatch, re - q0);
    __syncthreads();
    if (tid < nb) pl[tid] = xpos[q0 + tid];
    gno_stage_rows(zl, z, xpos, q0, nb, kdim, kdP, tid);
    gno_stage_rows(dml, dm, xpos, q0, nb, cout, cP, tid);
    __syncthreads();
    if (dz && kk < kdim) {
      for (int e0 = og; e0 < nb; e0 += 4 * OG) {    // register tile: 4 edges share every T read
        int er[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) er[i] = min(e0 + i * OG, nb - 1);
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int o = 0; o < cP; o += 4) {
          const float t0 = Tl[kk * ts + o], t1 = Tl[kk * ts + o + 1], t2 = Tl[kk * ts + o + 2], t3 = Tl[kk * ts + o + 3];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 d4 = *reinterpret_cast<const float4 *>(&dml[er[i] * cP + o]);
            s[i] = fmaf(t0, d4.x, fmaf(t1, d4.y, fmaf(t2, d4.z, fmaf(t3, d4.w, s[i]))));
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (e0 + i * OG < nb) dz[(size_t)pl[e0 + i * OG] * kdim + kk] = s[i];
      }
    }
    if (kk < kdim && ob < cP) {
      for (int e2 = 0; e2 < nb; ++e2) {
        const float zk = zl[e2 * kdP + kk];
        const float *dr = dml + e2 * cP + ob;
#pragma unroll
        for (int a = 0; a < MAXA; a += 4) {
          if (a < A && ob + a < cP) {
            const float4 d4 = *reinterpret_cast<const float4 *>(dr + a);
            acc[a] = fmaf(d4.x, zk, acc[a]);
            acc[a + 1] = fmaf(d4.y, zk, acc[a + 1]);
            acc[a + 2] = fmaf(d4.z, zk, acc[a + 2]);
            acc[a + 3] = fmaf(d4.w, zk, acc[a + 3]);
          }
        }
      }
    }
    if (tid < cout)
      for (int e2 = 0; e2 < nb; ++e2) accb += dml[e2 * cP + tid];
  }
  if (dT && kk < kdim) {
#pragma unroll
    for (int a = 0; a < MAXA; ++a) {
      const int o = ob + a;
      if (a < A && o < cout) dT[(size_t)j * total + (size_t)o * kdim + kk] = acc[a];
    }
  }
  if (dBh && tid < cout) dBh[(size_t)j * cout + tid] = accb;
}

// ---- GAT-style attention over incoming edges [GraphNeuralNetworks.jl GATConv] --------------------------------
// per (target i, head k): logit_p = leakyrelu(al[i][k] + ar[s_p][k]); alpha = softmax over the row (max-subtracted);
// out[i][k*C + c] = sum_p alpha_p Wx[s_p][k*C + c].  One wave per target row, loops over heads.
__global__ __launch_bounds__(256) void gat_fwd_kernel(int n_nodes, int heads, int c, float slope, const int *__restrict__ rowptr,
                                                      const int *__restrict__ col, const float *__restrict__ wx,
                                                      const float *__restrict__ al, const float *__restrict__ ar,
                                                      float *__restrict__ out, float *__restrict__ alpha) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  const int hc = heads * c;
  for (int k = 0; k < heads; ++k) {
    const float ali = al[(size_t)row * heads + k];
    float mx = -INFINITY;
    for (int p = rs + lane; p < re; p += 64) {
      float v = ali + ar[(size_t)col[p] * heads + k];
      v = v > 0.f ? v : slope * v;
      mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
    for (int p = rs + lane; p < re; p += 64) {
      float v = ali + ar[(size_t)col[p] * heads + k];
      v = v > 0.f ? v : slope * v;
      const float e = expf(v - mx);
      alpha[(size_t)p * heads + k] = e;
      sum += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv = 1.0f / sum;
    for (int p = rs + lane; p < re; p += 64) alpha[(size_t)p * heads + k] *= inv;
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  for (int f = lane; f < hc; f += 64) {
    const int k = f / c;
    float acc = 0.f;
    for (int p = rs; p < re; ++p) acc = fmaf(alpha[(size_t)p * heads + k], wx[(size_t)col[p] * hc + f], acc);
    out[(size_t)row * hc + f] = acc;
  }
}

// Same layer, restructured for short rows (heads a power of two <= 16): one wave per target row, lanes laid out as
// (edge slot, head) so that a block of 64 / heads edges computes all its logits at once; max, sum and the attention
// coefficients stay in registers (cross-lane shuffles), alpha is written once (coalesced) for the pullback, and the
// aggregation reads each edge's coefficient with a shuffle and issues the block's Wx row loads before the first FMA.
// Summation order = edge order inside the row, as the reference's scatter.
template <int HEADS>
__global__ __launch_bounds__(256) void gat_fwd_blocked_kernel(int n_nodes, int c, float slope, const int *__restrict__ rowptr,
                                                              const int *__restrict__ col, const float *__restrict__ wx,
                                                              const float *__restrict__ al, const float *__restrict__ ar,
                                                              float *__restrict__ out, float *__restrict__ alpha) {
  constexpr int EPB = 64 / HEADS;                    // edges per block
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  const int hc = HEADS * c;
  const int ke = lane % HEADS, le = lane / HEADS;
  const float ali = al[(size_t)row * HEADS + ke];
  auto score = [&](int pp, int &cj) {
    cj = col[pp];
    const float v = ali + ar[(size_t)cj * HEADS + ke];
    return v > 0.f ? v : slope * v;
  };
  auto over_edges_max = [&](float v) {
#pragma unroll
    for (int o = HEADS; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
  };
  auto over_edges_sum = [&](float v) {
#pragma unroll
    for (int o = HEADS; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
  };
  const bool one_block = re - rs <= EPB;              // wave-uniform: the whole row in one block => ONE gather chain
  // pass 1: row maximum per head; pass 2: denominator (a one-block row keeps its scores in registers)
  float mx = -INFINITY, v0 = -INFINITY;
  int cj0 = 0;
  if (one_block) {
    if (rs + le < re) v0 = score(rs + le, cj0);
    mx = v0;
  } else {
    for (int p0 = rs; p0 < re; p0 += EPB) {
      int cj;
      if (p0 + le < re) mx = fmaxf(mx, score(p0 + le, cj));
    }
  }
  mx = over_edges_max(mx);
  float sum = 0.f, e0 = 0.f;
  if (one_block) {
    if (rs + le < re) e0 = fast_exp(v0 - mx);
    sum = e0;
  } else {
    for (int p0 = rs; p0 < re; p0 += EPB) {
      int cj;
      if (p0 + le < re) sum += fast_exp(score(p0 + le, cj) - mx);
    }
  }
  sum = over_edges_sum(sum);
  const float inv = fast_rcp(sum);
  // pass 3: coefficients of a block in registers, aggregation of the block's rows
  float acc[4] = {0.f, 0.f, 0.f, 0.f};               // features lane, lane + 64, ... (hc <= 256)
  for (int p0 = rs; p0 < re; p0 += EPB) {
    const int nb = min(EPB, re - p0);
    int cj = cj0;
    float a = 0.f;
    if (le < nb) {
      a = (one_block ? e0 : fast_exp(score(p0 + le, cj) - mx)) * inv;
      alpha[(size_t)(p0 + le) * HEADS + ke] = a;       // = alpha[rs * HEADS + lane ...]: contiguous over the wave
    }
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
      if (64 * fb >= hc) break;                        // wave-uniform
      const bool fok = lane + 64 * fb < hc;            // no divergence around the shuffles: every lane is a shuffle source
      const int f = fok ? lane + 64 * fb : 0;
      const int kf = f / c;
      float w[EPB];
#pragma unroll
      for (int e = 0; e < EPB; ++e) {
        const int ce = __shfl(cj, e * HEADS);
        w[e] = (e < nb && fok) ? wx[(size_t)ce * hc + f] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < EPB; ++e) {
        const float ae = __shfl(a, e * HEADS + kf);
        acc[fb] = fmaf(ae, w[e], acc[fb]);             // ae = 0 past the block's last edge
      }
    }
  }
#pragma unroll
  for (int fb = 0; fb < 4; ++fb) {
    const int f = lane + 64 * fb;
    if (f < hc) out[(size_t)row * hc + f] = acc[fb];
  }
}

// backward, target side: dalpha_p = <dout[i], Wx[s_p]>_head; dlogit = alpha (dalpha - sum alpha dalpha);
// dscore_p = dlogit * leakyrelu'(score); dal[i] = sum_p dscore_p.  Stores dscore (for the source side).
__global__ __launch_bounds__(256) void gat_bwd_target_kernel(int n_nodes, int heads, int c, float slope,
                                                             const int *__restrict__ rowptr, const int *__restrict__ col,
                                                             const float *__restrict__ wx, const float *__restrict__ al,
                                                             const float *__restrict__ ar, const float *__restrict__ alpha,
                                                             const float *__restrict__ dout, float *__restrict__ dscore,
                                                             float *__restrict__ dal) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  const int hc = heads * c;
  for (int k = 0; k < heads; ++k) {
    // dalpha for every edge of the row (lanes over edges), then the softmax pullback
    float dot_sum = 0.f;
    for (int p = rs + lane; p < re; p += 64) {
      float da = 0.f;
      for (int cc = 0; cc < c; ++cc) da = fmaf(dout[(size_t)row * hc + k * c + cc], wx[(size_t)col[p] * hc + k * c + cc], da);
      dscore[(size_t)p * heads + k] = da;
      dot_sum = fmaf(alpha[(size_t)p * heads + k], da, dot_sum);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot_sum += __shfl_xor(dot_sum, off);
    const float ali = al[(size_t)row * heads + k];
    float s = 0.f;
    for (int p = rs + lane; p < re; p += 64) {
      const float a = alpha[(size_t)p * heads + k];
      const float dl = a * (dscore[(size_t)p * heads + k] - dot_sum);
      const float sc = ali + ar[(size_t)col[p] * heads + k];
      const float g = dl * (sc > 0.f ? 1.0f : slope);
      dscore[(size_t)p * heads + k] = g;
      s += g;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) dal[(size_t)row * heads + k] = s;
  }
}

// backward, source side: dWx[j] = sum_{p leaving j} alpha_p dout[t_p];  dar[j] = sum_p dscore_p
__global__ __launch_bounds__(256) void gat_bwd_source_kernel(int n_nodes, int heads, int c, const int *__restrict__ rowptr_s,
                                                             const int *__restrict__ col_s, const int *__restrict__ xpos,
                                                             const float *__restrict__ alpha, const float *__restrict__ dout,
                                                             const float *__restrict__ dscore, float *__restrict__ dwx,
                                                             float *__restrict__ dar) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr_s[row], re = rowptr_s[row + 1];
  const int hc = heads * c;
  for (int f = lane; f < hc; f += 64) {
    const int k = f / c;
    float acc = 0.f;
    for (int q = rs; q < re; ++q) acc = fmaf(alpha[(size_t)xpos[q] * heads + k], dout[(size_t)col_s[q] * hc + f], acc);
    dwx[(size_t)row * hc + f] = acc;
  }
  for (int k = lane; k < heads; k += 64) {
    float s = 0.f;
    for (int q = rs; q < re; ++q) s += dscore[(size_t)xpos[q] * heads + k];
    dar[(size_t)row * heads + k] = s;
  }
}

// Blocked forms of the two pullback kernels (heads a power of two <= 16, c a multiple of 4): lanes = (edge slot, head).
// Target side: every (edge, head) dot product <dout_i, Wx_j>_head is one lane's c/4 16-byte loads; the softmax pullback
// (row sum of alpha * dalpha per head) and dal are cross-lane sums; dscore is written once, coalesced.
template <int HEADS>
__global__ __launch_bounds__(256) void gat_bwd_target_blocked_kernel(int n_nodes, int c, float slope, const int *__restrict__ rowptr,
                                                                     const int *__restrict__ col, const float *__restrict__ wx,
                                                                     const float *__restrict__ al, const float *__restrict__ ar,
                                                                     const float *__restrict__ alpha, const float *__restrict__ dout,
                                                                     float *__restrict__ dscore, float *__restrict__ dal) {
  constexpr int EPB = 64 / HEADS;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  const int hc = HEADS * c;
  const int ke = lane % HEADS, le = lane / HEADS;
  const float ali = al[(size_t)row * HEADS + ke];
  const float4 *d4 = reinterpret_cast<const float4 *>(dout + (size_t)row * hc + ke * c);
  auto over_edges_sum = [&](float v) {
#pragma unroll
    for (int o = HEADS; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
  };
  auto dalpha = [&](int pp) {            // <dout[row], Wx[col_pp]> over this lane's head
    const float4 *w4 = reinterpret_cast<const float4 *>(wx + (size_t)col[pp] * hc + ke * c);
    float da = 0.f;
    for (int cc = 0; cc < c / 4; ++cc) {
      const float4 a = d4[cc], b = w4[cc];
      da = fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, fmaf(a.w, b.w, da))));
    }
    return da;
  };
  // pass 1: sum_p alpha_p dalpha_p per head
  float dot_sum = 0.f;
  for (int p0 = rs; p0 < re; p0 += EPB)
    if (p0 + le < re) dot_sum = fmaf(alpha[(size_t)(p0 + le) * HEADS + ke], dalpha(p0 + le), dot_sum);
  dot_sum = over_edges_sum(dot_sum);
  // pass 2: dscore, dal  (rows of up to 64 / HEADS edges recompute nothing: the loop body runs once per pass)
  float s = 0.f;
  for (int p0 = rs; p0 < re; p0 += EPB) {
    if (p0 + le < re) {
      const int pp = p0 + le;
      const float dl = alpha[(size_t)pp * HEADS + ke] * (dalpha(pp) - dot_sum);
      const float sc = ali + ar[(size_t)col[pp] * HEADS + ke];
      const float g = dl * (sc > 0.f ? 1.0f : slope);
      dscore[(size_t)pp * HEADS + ke] = g;
      s += g;
    }
  }
  s = over_edges_sum(s);
  if (le == 0) dal[(size_t)row * HEADS + ke] = s;
}

// Source side: dWx[j] = sum_{p leaving j} alpha_p dout[t_p];  dar[j] = sum_p dscore_p, the block's loads issued before
// the first FMA, coefficients fetched once per (edge, head) and broadcast with shuffles.
template <int HEADS>
__global__ __launch_bounds__(256) void gat_bwd_source_blocked_kernel(int n_nodes, int c, const int *__restrict__ rowptr_s,
                                                                     const int *__restrict__ col_s, const int *__restrict__ xpos,
                                                                     const float *__restrict__ alpha, const float *__restrict__ dout,
                                                                     const float *__restrict__ dscore, float *__restrict__ dwx,
                                                                     float *__restrict__ dar) {
  constexpr int EPB = 64 / HEADS;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_nodes) return;
  const int rs = rowptr_s[row], re = rowptr_s[row + 1];
  const int hc = HEADS * c;
  const int ke = lane % HEADS, le = lane / HEADS;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float sdar = 0.f;
  for (int q0 = rs; q0 < re; q0 += EPB) {
    const int nb = min(EPB, re - q0);
    int ct = 0;
    float a = 0.f;
    if (le < nb) {
      const int pp = xpos[q0 + le];
      ct = col_s[q0 + le];
      a = alpha[(size_t)pp * HEADS + ke];
      sdar += dscore[(size_t)pp * HEADS + ke];
    }
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
      if (64 * fb >= hc) break;                        // wave-uniform
      const bool fok = lane + 64 * fb < hc;
      const int f = fok ? lane + 64 * fb : 0;
      const int kf = f / c;
      float w[EPB];
#pragma unroll
      for (int e = 0; e < EPB; ++e) {
        const int te = __shfl(ct, e * HEADS);
        w[e] = (e < nb && fok) ? dout[(size_t)te * hc + f] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < EPB; ++e) acc[fb] = fmaf(__shfl(a, e * HEADS + kf), w[e], acc[fb]);
    }
  }
#pragma unroll
  for (int fb = 0; fb < 4; ++fb)
    if (lane + 64 * fb < hc) dwx[(size_t)row * hc + lane + 64 * fb] = acc[fb];
#pragma unroll
  for (int o = HEADS; o < 64; o <<= 1) sdar += __shfl_xor(sdar, o);
  if (le == 0) dar[(size_t)row * HEADS + ke] = sdar;
}

// score halves: al[n][k] = sum_c a[c][k] Wx[n][k*C + c];  ar with a[C + c][k]   (a is (2C x H) column-major)
__global__ void gat_scores_kernel(int64_t n_nodes, int heads, int c, const float *__restrict__ wx, const float *__restrict__ a,
                                  float *__restrict__ al, float *__restrict__ ar) {
  const int64_t total = n_nodes * heads;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / heads;
    const int k = (int)(i % heads);
    float sl = 0.f, sr = 0.f;
    for (int cc = 0; cc < c; ++cc) {
      const float w = wx[n * heads * c + k * c + cc];
      sl = fmaf(a[(size_t)k * 2 * c + cc], w, sl);
      sr = fmaf(a[(size_t)k * 2 * c + c + cc], w, sr);
    }
    al[i] = sl;
    ar[i] = sr;
  }
}

// pullback of gat_scores, part 1: dWx[n][k*C+c] += a_l[c][k] dal[n][k] + a_r[c][k] dar[n][k]
__global__ void gat_scores_bwd_dwx_kernel(int64_t n_nodes, int heads, int c, const float *__restrict__ a,
                                          const float *__restrict__ dal, const float *__restrict__ dar,
                                          float *__restrict__ dwx) {
  const int hc = heads * c;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes * hc; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / hc;
    const int f = (int)(i % hc), k = f / c, cc = f % c;
    dwx[i] += a[(size_t)k * 2 * c + cc] * dal[n * heads + k] + a[(size_t)k * 2 * c + c + cc] * dar[n * heads + k];
  }
}

// part 2: da[(which*C + c) + 2C*k] = sum_n Wx[n][k*C + c] * (which ? dar : dal)[n][k]; one block per element,
// fixed-order tree reduction (deterministic)
__global__ __launch_bounds__(256) void gat_scores_bwd_da_kernel(int64_t n_nodes, int heads, int c, const float *__restrict__ wx,
                                                                const float *__restrict__ dal, const float *__restrict__ dar,
                                                                float *__restrict__ da) {
  __shared__ float red[256];
  const int e = blockIdx.x;                 // 0 .. 2*c*heads - 1, laid out as a (2C x H) column-major matrix
  const int k = e / (2 * c), r = e % (2 * c), which = r / c, cc = r % c;
  const float *gsrc = which ? dar : dal;
  float s = 0.f;
  for (int64_t n = threadIdx.x; n < n_nodes; n += 256) s = fmaf(wx[n * heads * c + k * c + cc], gsrc[n * heads + k], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) da[e] = red[0];
}

// SpectralConv edge weights: w_e = 1/2 cos(n e / 2) cot(e / 2)   (src/layers.jl:654)
__global__ void spectral_weight_kernel(int64_t n_edges, float nn, const float *__restrict__ e, float *__restrict__ w) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_edges; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = e[i];
    w[i] = cosf(v * nn * 0.5f) * (cosf(v * 0.5f) / sinf(v * 0.5f)) * 0.5f;
  }
}

inline int blocks_for(int64_t count) { return (int)std::min<int64_t>((count + 255) / 256, 4096); }
inline unsigned rows4(int64_t rows) { return (unsigned)((rows + 3) / 4); }

}  // namespace

int32_t launch_dense_dz(int64_t count, int act, const float *dy, const float *z, float *dz, hipStream_t stream) {
  if (count == 0) return NGPDE_OK;
  hipLaunchKernelGGL(dense_dz_kernel, dim3(blocks_for(count)), dim3(256), 0, stream, count, act, dy, z, dz);
  NGPDE_LAUNCH_CHECK("dense_dz_kernel");
  return NGPDE_OK;
}

int32_t launch_edge_permute(const ngpde_graph *g, int d, bool inverse, const float *src, float *dst, hipStream_t stream) {
  if (g->n_edges == 0 || d == 0) return NGPDE_OK;
  hipLaunchKernelGGL(edge_permute_kernel, dim3(blocks_for(g->n_edges * d)), dim3(256), 0, stream, g->n_edges, d, g->by_t.eid,
                     inverse ? 1 : 0, src, dst);
  NGPDE_LAUNCH_CHECK("edge_permute_kernel");
  return NGPDE_OK;
}

int32_t launch_edge_combine_fwd(const ngpde_graph *g, int h, int act, const float *P, const float *Q, const float *Eterm,
                                float *a_out, float *z_out, hipStream_t stream) {
  if (g->n_nodes == 0 || h == 0) return NGPDE_OK;
  if (h % 4 == 0 && h <= 256 && al16(P) && al16(Q) && al16(Eterm) && al16(a_out) && al16(z_out)) {
    NGPDE_DPL_DISPATCH(edge_combine_fwd4_kernel, h / 4, dim3(rows4(g->n_nodes)), stream, (int)g->n_nodes, h / 4, act, g->by_t.rowptr,
                       g->by_t.col, reinterpret_cast<const float4 *>(P), reinterpret_cast<const float4 *>(Q),
                       reinterpret_cast<const float4 *>(Eterm), reinterpret_cast<float4 *>(a_out), reinterpret_cast<float4 *>(z_out))
    NGPDE_LAUNCH_CHECK("edge_combine_fwd4_kernel");
    return NGPDE_OK;
  }
  hipLaunchKernelGGL(edge_combine_fwd_kernel, dim3(rows4(g->n_nodes)), dim3(256), 0, stream, (int)g->n_nodes, h, act,
                     g->by_t.rowptr, g->by_t.col, P, Q, Eterm, a_out, z_out);
  NGPDE_LAUNCH_CHECK("edge_combine_fwd_kernel");
  return NGPDE_OK;
}

int32_t launch_edge_combine_bwd(const ngpde_graph *g, int h, int act, const float *da, const float *z, float *dz, float *dP,
                                float *dQ, hipStream_t stream) {
  if (g->n_nodes == 0 || h == 0) return NGPDE_OK;
  if (h % 4 == 0 && h <= 256 && al16(da) && al16(z) && al16(dz) && al16(dP)) {
    NGPDE_DPL_DISPATCH(edge_combine_bwd_target4_kernel, h / 4, dim3(rows4(g->n_nodes)), stream, (int)g->n_nodes, h / 4, act,
                       g->by_t.rowptr, reinterpret_cast<const float4 *>(da), reinterpret_cast<const float4 *>(z),
                       reinterpret_cast<float4 *>(dz), reinterpret_cast<float4 *>(dP))
  } else {
    hipLaunchKernelGGL(edge_combine_bwd_target_kernel, dim3(rows4(g->n_nodes)), dim3(256), 0, stream, (int)g->n_nodes, h, act,
                       g->by_t.rowptr, da, z, dz, dP);
  }
  NGPDE_LAUNCH_CHECK("edge_combine_bwd_target_kernel");
  if (dQ) {
    hipLaunchKernelGGL(edge_sum_by_source_kernel, dim3(rows4(g->n_nodes)), dim3(256), 0, stream, (int)g->n_nodes, h,
                       g->by_s.rowptr, g->by_s.xpos, dz, dQ);
    NGPDE_LAUNCH_CHECK("edge_sum_by_source_kernel");
  }
  return NGPDE_OK;
}

int32_t launch_edge_sum_by_source(const ngpde_graph *g, int h, const float *per_edge, float *out, hipStream_t stream) {
  if (g->n_nodes == 0 || h == 0) return NGPDE_OK;
  hipLaunchKernelGGL(edge_sum_by_source_kernel, dim3(rows4(g->n_nodes)), dim3(256), 0, stream, (int)g->n_nodes, h,
                     g->by_s.rowptr, g->by_s.xpos, per_edge, out);
  NGPDE_LAUNCH_CHECK("edge_sum_by_source_kernel");
  return NGPDE_OK;
}

int32_t launch_segment_reduce_fwd(const ngpde_graph *g, int d, int aggr, const float *M, float *out, hipStream_t stream) {
  if (g->n_nodes == 0 || d == 0) return NGPDE_OK;
  if ((aggr == NGPDE_AGGR_SUM || aggr == NGPDE_AGGR_MEAN) && d % 4 == 0 && d <= 256 && al16(M) && al16(out)) {
    NGPDE_DPL_DISPATCH(segment_sum4_kernel, d / 4, dim3(rows4(g->n_nodes)), stream, (int)g->n_nodes, d / 4, aggr == NGPDE_AGGR_MEAN ? 1 : 0,
                       g->by_t.rowptr, reinterpret_cast<const float4 *>(M), reinterpret_cast<float4 *>(out))
    NGPDE_LAUNCH_CHECK("segment_sum4_kernel");
    return NGPDE_OK;
  }
  hipLaunchKernelGGL(segment_reduce_fwd_kernel, dim3(rows4(g->n_nodes)), dim3(256), 0, stream, (int)g->n_nodes, d, aggr,
                     g->by_t.rowptr, M, out);
  NGPDE_LAUNCH_CHECK("segment_reduce_fwd_kernel");
  return NGPDE_OK;
}

int32_t launch_segment_reduce_bwd(const ngpde_graph *g, int d, int aggr, const float *M, const float *out, const float *dout,
                                  float *dM, hipStream_t stream) {
  if (g->n_nodes == 0 || d == 0) return NGPDE_OK;
  hipLaunchKernelGGL(segment_reduce_bwd_kernel, dim3(rows4(g->n_nodes)), dim3(256), 0, stream, (int)g->n_nodes, d, aggr,
                     g->by_t.rowptr, M, out, dout, dM);
  NGPDE_LAUNCH_CHECK("segment_reduce_bwd_kernel");
  return NGPDE_OK;
}

int32_t launch_gno_contract_fwd(const ngpde_graph *g, int cin, int cout, const float *K, const float *h, float *m,
                                hipStream_t stream) {
  if (g->n_edges == 0) return NGPDE_OK;
  hipLaunchKernelGGL(gno_contract_fwd_kernel, dim3(rows4(g->n_edges)), dim3(256), 0, stream, g->n_edges, cin, cout,
                     g->by_t.col, K, h, m);
  NGPDE_LAUNCH_CHECK("gno_contract_fwd_kernel");
  return NGPDE_OK;
}

int32_t launch_gno_contract_bwd(const ngpde_graph *g, int cin, int cout, const float *K, const float *h, const float *dm,
                                float *dK, float *dhe, hipStream_t stream) {
  if (g->n_edges == 0) return NGPDE_OK;
  hipLaunchKernelGGL(gno_contract_bwd_kernel, dim3(rows4(g->n_edges)), dim3(256), 0, stream, g->n_edges, cin, cout,
                     g->by_t.col, K, h, dm, dK, dhe);
  NGPDE_LAUNCH_CHECK("gno_contract_bwd_kernel");
  return NGPDE_OK;
}

static int gno_kp_log2(int kdim) {
  int l = 0;
  while ((1 << l) < kdim) ++l;
  return l;
}

static int gno_pad4(int v) { return (v + 3) & ~3; }

// rows of dT_j per thread in the pullback: ceil(pad4(cout) / OG) rounded up to a multiple of 4
static int gno_rows_per_thread(int cout, int kdim) {
  const int og = 256 >> gno_kp_log2(kdim);
  return gno_pad4((gno_pad4(cout) + og - 1) / og);
}

static size_t gno_lds_bytes(int cout, int kdim, bool bwd) {
  const int cP = gno_pad4(cout), kdP = gno_pad4(kdim);
  return ((size_t)kdP * (cP + 1) + (size_t)kGnoBatch * kdP + (bwd ? (size_t)kGnoBatch * cP : 0)) * sizeof(float);
}

bool gno_apply_supported(int cout, int kdim) {
  if (cout <= 0 || kdim <= 0 || cout > 256 || kdim > 256) return false;
  return gno_rows_per_thread(cout, kdim) <= 32 && gno_lds_bytes(cout, kdim, true) <= 60 * 1024;
}

int32_t launch_gno_apply_fwd(const ngpde_graph *g, int cout, int kdim, const float *T, const float *Bh, const float *z,
                             float *m, hipStream_t stream) {
  if (g->n_edges == 0) return NGPDE_OK;
  if (gno_apply_mfma_supported(cout, kdim)) return launch_gno_apply_mfma_fwd(g, cout, kdim, T, Bh, z, m, stream);
  hipLaunchKernelGGL(gno_apply_fwd_kernel, dim3((unsigned)g->n_nodes), dim3(256), gno_lds_bytes(cout, kdim, false), stream,
                     (int)g->n_nodes, cout, kdim, g->by_s.rowptr, g->by_s.xpos, T, Bh, z, m);
  NGPDE_LAUNCH_CHECK("gno_apply_fwd_kernel");
  return NGPDE_OK;
}

int32_t launch_gno_apply_bwd(const ngpde_graph *g, int cout, int kdim, const float *T, const float *z, const float *dm,
                             float *dT, float *dBh, float *dz, hipStream_t stream) {
  if (g->n_nodes == 0) return NGPDE_OK;
  if (gno_apply_mfma_supported(cout, kdim)) return launch_gno_apply_mfma_bwd(g, cout, kdim, T, z, dm, dT, dBh, dz, stream);
  hipLaunchKernelGGL(gno_apply_bwd_kernel, dim3((unsigned)g->n_nodes), dim3(256), gno_lds_bytes(cout, kdim, true), stream,
                     (int)g->n_nodes, cout, kdim, gno_kp_log2(kdim), gno_rows_per_thread(cout, kdim), g->by_s.rowptr,
                     g->by_s.xpos, T, z, dm, dT, dBh, dz);
  NGPDE_LAUNCH_CHECK("gno_apply_bwd_kernel");
  return NGPDE_OK;
}

int32_t launch_gat_scores(int64_t n, int heads, int c, const float *wx, const float *a, float *al, float *ar,
                          hipStream_t stream) {
  if (n == 0) return NGPDE_OK;
  hipLaunchKernelGGL(gat_scores_kernel, dim3(blocks_for(n * heads)), dim3(256), 0, stream, n, heads, c, wx, a, al, ar);
  NGPDE_LAUNCH_CHECK("gat_scores_kernel");
  return NGPDE_OK;
}

int32_t launch_gat_fwd(const ngpde_graph *g, int heads, int c, float slope, const float *wx, const float *al, const float *ar,
                       float *out, float *alpha, hipStream_t stream) {
  if (g->n_nodes == 0) return NGPDE_OK;
  const bool no_fused = getenv("NGPDE_NO_FUSED_GAT") != nullptr;   // read per call: tests switch it at run time
  if (!no_fused && gat_fused_supported(g, heads, c)) return launch_gat_fused_fwd(g, heads, c, slope, wx, al, ar, out, alpha, stream);
  const dim3 grid(rows4(g->n_nodes)), block(256);
  const int n = (int)g->n_nodes;
  if (heads * c <= 256 && (heads == 1 || heads == 2 || heads == 4 || heads == 8 || heads == 16)) {
    switch (heads) {
      case 1: hipLaunchKernelGGL(gat_fwd_blocked_kernel<1>, grid, block, 0, stream, n, c, slope, g->by_t.rowptr, g->by_t.col, wx, al, ar, out, alpha); break;
      case 2: hipLaunchKernelGGL(gat_fwd_blocked_kernel<2>, grid, block, 0, stream, n, c, slope, g->by_t.rowptr, g->by_t.col, wx, al, ar, out, alpha); break;
      case 4: hipLaunchKernelGGL(gat_fwd_blocked_kernel<4>, grid, block, 0, stream, n, c, slope, g->by_t.rowptr, g->by_t.col, wx, al, ar, out, alpha); break;
      case 8: hipLaunchKernelGGL(gat_fwd_blocked_kernel<8>, grid, block, 0, stream, n, c, slope, g->by_t.rowptr, g->by_t.col, wx, al, ar, out, alpha); break;
      default: hipLaunchKernelGGL(gat_fwd_blocked_kernel<16>, grid, block, 0, stream, n, c, slope, g->by_t.rowptr, g->by_t.col, wx, al, ar, out, alpha); break;
    }
    NGPDE_LAUNCH_CHECK("gat_fwd_blocked_kernel");
    return NGPDE_OK;
  }
  hipLaunchKernelGGL(gat_fwd_kernel, grid, block, 0, stream, n, heads, c, slope, g->by_t.rowptr, g->by_t.col, wx, al, ar, out, alpha);
  NGPDE_LAUNCH_CHECK("gat_fwd_kernel");
  return NGPDE_OK;
}

int32_t launch_gat_bwd(const ngpde_graph *g, int heads, int c, float slope, const float *wx, const float *a, const float *al,
                       const float *ar, const float *alpha, const float *dout, float *dscore, float *dal, float *dar,
                       float *dwx, float *da, hipStream_t stream) {
  if (g->n_nodes == 0) return NGPDE_OK;
  const int n = (int)g->n_nodes;
  const bool blocked = heads * c <= 256 && c % 4 == 0 && (heads == 1 || heads == 2 || heads == 4 || heads == 8 || heads == 16) &&
                       ((reinterpret_cast<uintptr_t>(wx) | reinterpret_cast<uintptr_t>(dout)) & 15) == 0;
  if (blocked) {
#define NGPDE_GAT_BWD(HH)                                                                                                  \
  hipLaunchKernelGGL(gat_bwd_target_blocked_kernel<HH>, dim3(rows4(n)), dim3(256), 0, stream, n, c, slope, g->by_t.rowptr, \
                     g->by_t.col, wx, al, ar, alpha, dout, dscore, dal);                                                     \
  hipLaunchKernelGGL(gat_bwd_source_blocked_kernel<HH>, dim3(rows4(n)), dim3(256), 0, stream, n, c, g->by_s.rowptr,        \
                     g->by_s.col, g->by_s.xpos, alpha, dout, dscore, dwx, dar);
    switch (heads) {
      case 1: NGPDE_GAT_BWD(1) break;
      case 2: NGPDE_GAT_BWD(2) break;
      case 4: NGPDE_GAT_BWD(4) break;
      case 8: NGPDE_GAT_BWD(8) break;
      default: NGPDE_GAT_BWD(16) break;
    }
#undef NGPDE_GAT_BWD
    NGPDE_LAUNCH_CHECK("gat_bwd_*_blocked_kernel");
  } else {
    hipLaunchKernelGGL(gat_bwd_target_kernel, dim3(rows4(n)), dim3(256), 0, stream, n, heads, c, slope, g->by_t.rowptr,
                       g->by_t.col, wx, al, ar, alpha, dout, dscore, dal);
    NGPDE_LAUNCH_CHECK("gat_bwd_target_kernel");
    hipLaunchKernelGGL(gat_bwd_source_kernel, dim3(rows4(n)), dim3(256), 0, stream, n, heads, c, g->by_s.rowptr, g->by_s.col,
                       g->by_s.xpos, alpha, dout, dscore, dwx, dar);
    NGPDE_LAUNCH_CHECK("gat_bwd_source_kernel");
  }
  hipLaunchKernelGGL(gat_scores_bwd_dwx_kernel, dim3(blocks_for((int64_t)n * heads * c)), dim3(256), 0, stream, (int64_t)n,
                     heads, c, a, dal, dar, dwx);
  NGPDE_LAUNCH_CHECK("gat_scores_bwd_dwx_kernel");
  hipLaunchKernelGGL(gat_scores_bwd_da_kernel, dim3(2 * c * heads), dim3(256), 0, stream, (int64_t)n, heads, c, wx, dal, dar,
                     da);
  NGPDE_LAUNCH_CHECK("gat_scores_bwd_da_kernel");
  return NGPDE_OK;
}

int32_t launch_spectral_weights(int64_t n_edges, float nn, const float *e, float *w, hipStream_t stream) {
  if (n_edges == 0) return NGPDE_OK;
  hipLaunchKernelGGL(spectral_weight_kernel, dim3(blocks_for(n_edges)), dim3(256), 0, stream, n_edges, nn, e, w);
  NGPDE_LAUNCH_CHECK("spectral_weight_kernel");
  return NGPDE_OK;
}

}  // namespace ngpde

// gcn_kernels.hip -- gfx950 kernels of the GCNConv hot path (/root/reference/src/layers.jl:200-239):
//   fused forward   : CSR segmented aggregation -> LDS tile -> fp32 MFMA (x W) -> bias/activation
//                     [-> Runge-Kutta stage combination], one launch per layer evaluation;
//   fused backward  : [A^T aggregation of the incoming gradient ->] [adjoint stage combination ->]
//                     act' mask -> fp32 MFMA (dZ x W^T and X3^T x dZ) -> per-block dW/db slabs;
//   generic kernels : any feature width (aggregation, dense, dense backward) for shapes outside the
//                     fused path.  No atomics anywhere: every output row / slab has one writer.
#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

constexpr int kThreads = 256;
constexpr int kTM = 32;  // node rows per workgroup: N=16384 -> 512 workgroups = 2 per CU, 2 waves per SIMD

template <int D>
struct Geo {
  static constexpr int LPR = D / 4;                          // lanes per feature row (float4 each)
  static constexpr int GROUPS = kThreads / LPR;              // row groups per workgroup
  static constexpr int R = (kTM + GROUPS - 1) / GROUPS;      // rows per group
  static constexpr int U = (R >= 4) ? 2 : 4;                 // edge unroll (rows in flight per lane = R*U)
  static constexpr int TS = D + 4;                           // LDS tile row stride (floats), keeps b128 alignment
  static constexpr int WS = D + 16;                          // LDS weight row stride (floats)
  static constexpr int W4 = (D * D / 4 + kThreads - 1) / kThreads;  // float4 of W per thread
  static constexpr int RT = kTM / 16;                        // 16-row MFMA tiles per workgroup
  static constexpr int CT = D / 16;                          // 16-col MFMA tiles
  static constexpr int WAVES = kThreads / 64;
  static constexpr int CGRP = WAVES / RT;                    // waves sharing one row tile
  static constexpr int CPW = (CT + CGRP - 1) / CGRP;         // column tiles per wave
};

struct CombDev {
  int n;
  const float *ptr[8];
  float coef[8];
  float coef_self;
};

__device__ __forceinline__ float4 comb_eval(const CombDev &c, float4 self, size_t idx4) {
  float4 v = f4_scale(c.coef_self, self);
  for (int k = 0; k < c.n; ++k) v = f4_fma(c.coef[k], reinterpret_cast<const float4 *>(c.ptr[k])[idx4], v);
  return v;
}

// Z tile [kTM][D] = A tile [kTM][D] (LDS, stride TS) x B [D][D] (LDS, stride WS), result to LDS (stride TS).
template <int D>
__device__ __forceinline__ void mfma_rows_times_b(const float *ldsA, const float *ldsB, float *ldsOut, int wave,
                                                  int lane) {
  using G = Geo<D>;
  const int rt = wave % G::RT;
  const int cg = wave / G::RT;
  const int i = lane & 15, kq = lane >> 4;
  f32x4 acc[G::CPW];
#pragma unroll
  for (int m = 0; m < G::CPW; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int kb = 0; kb < D / 16; ++kb) {
    const float4 a4 = *reinterpret_cast<const float4 *>(&ldsA[(rt * 16 + i) * G::TS + kb * 16 + 4 * kq]);
    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kb * 16 + 4 * kq + r;
#pragma unroll
      for (int m = 0; m < G::CPW; ++m) {
        const int ct = cg + G::CGRP * m;
        if (ct < G::CT) acc[m] = mfma16(av[r], ldsB[k * G::WS + ct * 16 + i], acc[m]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < G::CPW; ++m) {
    const int ct = cg + G::CGRP * m;
    if (ct < G::CT) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) ldsOut[(rt * 16 + 4 * kq + reg) * G::TS + ct * 16 + i] = acc[m][reg];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// fused forward:  y = act( (C (A+I) C x) Wt + b ),  optional RK stage combination on the fresh rows
// ---------------------------------------------------------------------------------------------------
struct FwdK {
  const float *x;
  const int *rowptr;
  const int2 *ent;
  const float *cnorm;
  int self_loops, n_nodes, act;
  const float *wt, *bias;
  float *y, *save_agg, *save_z;
  int has_comb;
  CombDev comb;
  float *comb_out;
};

template <int D>
__global__ __launch_bounds__(kThreads) void gcn_fused_fwd_kernel(const FwdK p) {
  using G = Geo<D>;
  __shared__ __attribute__((aligned(16))) float lds[kTM * G::TS * 2 + D * G::WS];
  float *ldsT = lds, *ldsZ = lds + kTM * G::TS, *ldsW = lds + 2 * kTM * G::TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = tid / G::LPR, q = tid % G::LPR;
  const int row0 = blockIdx.x * kTM;

  // W (16 KB at D=64, L2 resident) is requested first so it lands while the gather runs
  float4 wreg[G::W4];
#pragma unroll
  for (int k = 0; k < G::W4; ++k) {
    const int idx = tid + k * kThreads;
    wreg[k] = (idx < D * D / 4) ? reinterpret_cast<const float4 *>(p.wt)[idx] : f4_zero();
  }

  int rows[G::R];
  float4 acc[G::R];
  const bool active = grp * G::R < kTM;
#pragma unroll
  for (int r = 0; r < G::R; ++r) rows[r] = active ? row0 + grp * G::R + r : p.n_nodes;
  aggregate_rows<G::LPR, G::R, G::U>(reinterpret_cast<const float4 *>(p.x), p.rowptr, p.ent, p.cnorm,
                                     p.self_loops, p.n_nodes, rows, q, acc);
  if (active) {
#pragma unroll
    for (int r = 0; r < G::R; ++r) {
      *reinterpret_cast<float4 *>(&ldsT[(grp * G::R + r) * G::TS + 4 * q]) = acc[r];
      if (p.save_agg && rows[r] < p.n_nodes)
        reinterpret_cast<float4 *>(p.save_agg)[(size_t)rows[r] * G::LPR + q] = acc[r];
    }
  }
#pragma unroll
  for (int k = 0; k < G::W4; ++k) {
    const int idx = tid + k * kThreads;
    if (idx < D * D / 4) {
      const int kr = (idx * 4) / D, kc = (idx * 4) % D;
      *reinterpret_cast<float4 *>(&ldsW[kr * G::WS + kc]) = wreg[k];
    }
  }
  __syncthreads();
  mfma_rows_times_b<D>(ldsT, ldsW, ldsZ, wave, lane);
  __syncthreads();
  if (active) {
    const float4 b4 = p.bias ? reinterpret_cast<const float4 *>(p.bias)[q] : f4_zero();
#pragma unroll
    for (int r = 0; r < G::R; ++r) {
      if (rows[r] >= p.n_nodes) continue;
      const size_t idx4 = (size_t)rows[r] * G::LPR + q;
      float4 z = f4_add(*reinterpret_cast<const float4 *>(&ldsZ[(grp * G::R + r) * G::TS + 4 * q]), b4);
      if (p.save_z) reinterpret_cast<float4 *>(p.save_z)[idx4] = z;
      const float4 yv = f4_act(p.act, z);
      reinterpret_cast<float4 *>(p.y)[idx4] = yv;
      if (p.has_comb) reinterpret_cast<float4 *>(p.comb_out)[idx4] = comb_eval(p.comb, yv, idx4);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// fused backward of one layer evaluation
// ---------------------------------------------------------------------------------------------------
struct BwdK {
  const float *g_in;
  const int *rowptr;
  const int2 *ent;
  const float *cnorm;
  int self_loops, n_nodes, act;
  int has_comb;
  CombDev comb;
  float *store_t, *store_v;
  float v_scale;
  int do_dense;
  const float *z, *saved_agg, *wt;
  float *g_out, *slab_dw, *slab_db;
};

template <int D, bool AGG>
__global__ __launch_bounds__(kThreads) void gcn_fused_bwd_kernel(const BwdK p) {
  using G = Geo<D>;
  __shared__ __attribute__((aligned(16))) float lds[kTM * G::TS * 3 + D * G::WS];
  float *ldsDZ = lds, *ldsX = lds + kTM * G::TS, *ldsG = lds + 2 * kTM * G::TS, *ldsW = lds + 3 * kTM * G::TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = tid / G::LPR, q = tid % G::LPR;
  const int row0 = blockIdx.x * kTM;

  float4 wreg[G::W4];
  if (p.do_dense) {
#pragma unroll
    for (int k = 0; k < G::W4; ++k) {
      const int idx = tid + k * kThreads;
      wreg[k] = (idx < D * D / 4) ? reinterpret_cast<const float4 *>(p.wt)[idx] : f4_zero();
    }
  }

  int rows[G::R];
  float4 t[G::R];
  const bool active = grp * G::R < kTM;
#pragma unroll
  for (int r = 0; r < G::R; ++r) rows[r] = active ? row0 + grp * G::R + r : p.n_nodes;
  if (AGG) {
    aggregate_rows<G::LPR, G::R, G::U>(reinterpret_cast<const float4 *>(p.g_in), p.rowptr, p.ent, p.cnorm,
                                       p.self_loops, p.n_nodes, rows, q, t);
  } else {
#pragma unroll
    for (int r = 0; r < G::R; ++r)
      t[r] = rows[r] < p.n_nodes ? reinterpret_cast<const float4 *>(p.g_in)[(size_t)rows[r] * G::LPR + q] : f4_zero();
  }
  if (active) {
#pragma unroll
    for (int r = 0; r < G::R; ++r) {
      const bool ok = rows[r] < p.n_nodes;
      const size_t idx4 = (size_t)(ok ? rows[r] : 0) * G::LPR + q;
      float4 kbar = t[r];
      if (ok) {
        if (p.store_t) reinterpret_cast<float4 *>(p.store_t)[idx4] = t[r];
        if (p.has_comb) {
          float4 v = comb_eval(p.comb, t[r], idx4);
          if (p.store_v) reinterpret_cast<float4 *>(p.store_v)[idx4] = v;
          kbar = f4_scale(p.v_scale, v);
        }
      }
      if (p.do_dense) {
        float4 dz = f4_zero(), xa = f4_zero();
        if (ok) {
          dz = f4_mul(kbar, f4_dact(p.act, reinterpret_cast<const float4 *>(p.z)[idx4]));
          xa = reinterpret_cast<const float4 *>(p.saved_agg)[idx4];
        }
        *reinterpret_cast<float4 *>(&ldsDZ[(grp * G::R + r) * G::TS + 4 * q]) = dz;
        *reinterpret_cast<float4 *>(&ldsX[(grp * G::R + r) * G::TS + 4 * q]) = xa;
      }
    }
  }
  if (!p.do_dense) return;  // uniform for the whole grid
  // stage B = Wt^T : B[k = o][j = i] = Wt[i][o]
#pragma unroll
  for (int k = 0; k < G::W4; ++k) {
    const int idx = tid + k * kThreads;
    if (idx < D * D / 4) {
      const int wi = (idx * 4) / D, wo = (idx * 4) % D;
      ldsW[(wo + 0) * G::WS + wi] = wreg[k].x;
      ldsW[(wo + 1) * G::WS + wi] = wreg[k].y;
      ldsW[(wo + 2) * G::WS + wi] = wreg[k].z;
      ldsW[(wo + 3) * G::WS + wi] = wreg[k].w;
    }
  }
  __syncthreads();
  // G = dZ x Wt^T  (gradient w.r.t. the aggregated input)
  mfma_rows_times_b<D>(ldsDZ, ldsW, ldsG, wave, lane);
  // dWt[i][o] += sum_n X3[n][i] dZ[n][o]   (K = kTM rows of this tile)
  {
    const int i = lane & 15, kq = lane >> 4;
    constexpr int NT = G::CT * G::CT;
    float *slab = p.slab_dw + (size_t)blockIdx.x * D * D;
#pragma unroll 1
    for (int tt = wave; tt < NT; tt += G::WAVES) {
      const int mt = tt / G::CT, nt = tt % G::CT;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < kTM / 4; ++ks) {
        const float a = ldsX[(4 * ks + kq) * G::TS + mt * 16 + i];
        const float b = ldsDZ[(4 * ks + kq) * G::TS + nt * 16 + i];
        acc = mfma16(a, b, acc);
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float *dst = &slab[(mt * 16 + 4 * kq + reg) * D + nt * 16 + i];
        *dst += acc[reg];
      }
    }
    if (tid < D) {
      float s = 0.f;
#pragma unroll 8
      for (int n = 0; n < kTM; ++n) s += ldsDZ[n * G::TS + tid];
      p.slab_db[(size_t)blockIdx.x * D + tid] += s;
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int r = 0; r < G::R; ++r) {
      if (rows[r] >= p.n_nodes) continue;
      reinterpret_cast<float4 *>(p.g_out)[(size_t)rows[r] * G::LPR + q] =
          *reinterpret_cast<const float4 *>(&ldsG[(grp * G::R + r) * G::TS + 4 * q]);
    }
  }
}

__global__ void reduce_slabs_kernel(const float *__restrict__ slab, int n_slabs, int len, float *__restrict__ out) {
  // one workgroup of 256 threads per 64 output elements: 4 partial sums per element, then LDS combine
  __shared__ float part[4][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part_id = threadIdx.x >> 6;
  float s = 0.f;
  if (e < len)
    for (int b = part_id; b < n_slabs; b += 4) s += slab[(size_t)b * len + e];
  part[part_id][threadIdx.x & 63] = s;
  __syncthreads();
  if (part_id == 0 && e < len) out[e] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

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

CombDev to_dev(const Comb &c) {
  CombDev d;
  d.n = c.n;
  for (int k = 0; k < 8; ++k) {
    d.ptr[k] = c.ptr[k];
    d.coef[k] = c.coef[k];
  }
  d.coef_self = c.coef_self;
  return d;
}

#define NGPDE_LAUNCH_CHECK(name)                                                         \
  do {                                                                                   \
    hipError_t _e = hipGetLastError();                                                   \
    if (_e != hipSuccess) return fail(NGPDE_ERR_HIP, "%s launch failed: %s", name, hipGetErrorString(_e)); \
  } while (0)

}  // namespace

bool fused_supported(int din, int dout) { return din == dout && (din == 16 || din == 32 || din == 64 || din == 128); }
int fused_tile_rows() { return kTM; }
int fused_num_blocks(int64_t n_nodes) { return (int)((n_nodes + kTM - 1) / kTM); }

int32_t launch_fused_fwd(const FusedFwdArgs &a, hipStream_t stream) {
  const ngpde_graph *g = a.g;
  NGPDE_REQUIRE(g && g->has_norm, NGPDE_ERR_STATE, "GCN normalisation not set (call ngpde_graph_set_gcn_norm)");
  NGPDE_REQUIRE(fused_supported(a.d, a.d), NGPDE_ERR_UNSUPPORTED, "fused GCN path needs d in {16,32,64,128}, got %d", a.d);
  if (g->n_nodes == 0) return NGPDE_OK;
  FwdK k;
  k.x = a.x; k.rowptr = g->by_t.rowptr; k.ent = g->by_t.ent; k.cnorm = g->c;
  k.self_loops = g->self_loops; k.n_nodes = (int)g->n_nodes; k.act = a.act;
  k.wt = a.wt; k.bias = a.bias; k.y = a.y; k.save_agg = a.save_agg; k.save_z = a.save_z;
  k.has_comb = a.has_comb ? 1 : 0; k.comb = to_dev(a.comb); k.comb_out = a.comb_out;
  const dim3 grid(fused_num_blocks(g->n_nodes)), block(kThreads);
  switch (a.d) {
    case 16: hipLaunchKernelGGL(gcn_fused_fwd_kernel<16>, grid, block, 0, stream, k); break;
    case 32: hipLaunchKernelGGL(gcn_fused_fwd_kernel<32>, grid, block, 0, stream, k); break;
    case 64: hipLaunchKernelGGL(gcn_fused_fwd_kernel<64>, grid, block, 0, stream, k); break;
    default: hipLaunchKernelGGL(gcn_fused_fwd_kernel<128>, grid, block, 0, stream, k); break;
  }
  NGPDE_LAUNCH_CHECK("gcn_fused_fwd_kernel");
  return NGPDE_OK;
}

int32_t launch_fused_bwd(const FusedBwdArgs &a, hipStream_t stream) {
  const ngpde_graph *g = a.g;
  NGPDE_REQUIRE(g && g->has_norm, NGPDE_ERR_STATE, "GCN normalisation not set (call ngpde_graph_set_gcn_norm)");
  NGPDE_REQUIRE(fused_supported(a.d, a.d), NGPDE_ERR_UNSUPPORTED, "fused GCN path needs d in {16,32,64,128}, got %d", a.d);
  if (g->n_nodes == 0) return NGPDE_OK;
  BwdK k;
  k.g_in = a.g_in; k.rowptr = g->by_s.rowptr; k.ent = g->by_s.ent; k.cnorm = g->c;
  k.self_loops = g->self_loops; k.n_nodes = (int)g->n_nodes; k.act = a.act;
  k.has_comb = a.has_comb ? 1 : 0; k.comb = to_dev(a.comb);
  k.store_t = a.store_t; k.store_v = a.store_v; k.v_scale = a.v_scale;
  k.do_dense = a.do_dense ? 1 : 0; k.z = a.z; k.saved_agg = a.saved_agg; k.wt = a.wt;
  k.g_out = a.g_out; k.slab_dw = a.slab_dw; k.slab_db = a.slab_db;
  const dim3 grid(fused_num_blocks(g->n_nodes)), block(kThreads);
#define NGPDE_BWD_CASE(DD)                                                                         \
  case DD:                                                                                         \
    if (a.aggregate) hipLaunchKernelGGL((gcn_fused_bwd_kernel<DD, true>), grid, block, 0, stream, k);  \
    else hipLaunchKernelGGL((gcn_fused_bwd_kernel<DD, false>), grid, block, 0, stream, k);          \
    break;
  switch (a.d) {
    NGPDE_BWD_CASE(16)
    NGPDE_BWD_CASE(32)
    NGPDE_BWD_CASE(64)
    default:
      if (a.aggregate) hipLaunchKernelGGL((gcn_fused_bwd_kernel<128, true>), grid, block, 0, stream, k);
      else hipLaunchKernelGGL((gcn_fused_bwd_kernel<128, false>), grid, block, 0, stream, k);
      break;
  }
#undef NGPDE_BWD_CASE
  NGPDE_LAUNCH_CHECK("gcn_fused_bwd_kernel");
  return NGPDE_OK;
}

int32_t launch_reduce_slabs(const float *slab, int n_slabs, int len, float *out, hipStream_t stream) {
  if (len == 0) return NGPDE_OK;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((len + 63) / 64), dim3(256), 0, stream, slab, n_slabs, len, out);
  NGPDE_LAUNCH_CHECK("reduce_slabs_kernel");
  return NGPDE_OK;
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

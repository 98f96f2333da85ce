// gno_mfma.hip -- the reassociated GNOConv message (/root/reference/src/layers.jl:509-547) on the matrix pipe.
//   m_e = T_{s_e} z_e + Bh_{s_e},   T_j [out][k] = sum_i W2[o + out i][k] h_j[i]  (node level, ngpde_dense_forward)
// Grouped by SOURCE node the message is a real GEMM: the deg_out(j) edges leaving j share T_j,
//   M_j^T [deg][out] = Z_j [deg][k] x T_j^T [k][out]
// so one workgroup per source node multiplies 16-edge tiles of its gathered z rows with T_j on v_mfma_f32_16x16x4_f32 (exact
// fp32).  T_j never touches LDS in the forward: lane (o, kq) of an output-column tile reads its four consecutive k of row o
// straight from memory (16 bytes) and keeps them for all edge tiles.  BASELINE config 5 (128 => 128, k = 64, radius 0.1: 110
// edges per node): 7 edge tiles x 8 column tiles x 16 products per node.
// Pullback per source node:  dz_e = T_j^T dm_e  (DZ [deg][k] = DM [deg][out] x T_j),  dT_j = sum_e dm_e z_e^T  (DM^T x Z, the edge
// index contracted, accumulators in registers over the node's edge tiles),  dBh_j = sum_e dm_e.
// Used when out is a multiple of 16 (<= 256) and k is 16, 32 or 64; other shapes keep the register-tile kernels of
// mp_kernels.hip.  No atomics; every output row has one writer.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

constexpr int kET = 16;      // edges per MFMA tile
constexpr int kEB = 64;      // edges staged per pass of the forward (4 tiles)
constexpr int kEBb = 32;     // ... of the pullback (z and dm rows: 25 KB at 128 x 64)

// rows of a per-edge array ([E][w], p order) of the edges q0 .. q0 + nb of the source's list -> LDS [kEB][w + 4], zero rows
// beyond nb; 16-byte loads (w % 4 == 0)
template <int EB>
__device__ __forceinline__ void stage_edge_rows(float *dst, const float *__restrict__ src, const int *pl, int nb, int w, int tid) {
  const int w4 = w / 4, ls = w + 4;
  for (int idx = tid; idx < EB * w4; idx += 256) {
    const int e = idx / w4, c4 = idx - e * w4;
    const float4 v = e < nb ? reinterpret_cast<const float4 *>(src + (size_t)pl[e] * w)[c4] : f4_zero();
    *reinterpret_cast<float4 *>(&dst[e * ls + 4 * c4]) = v;
  }
}

// ---- forward ------------------------------------------------------------------------------------------------------------
// FUSED: the per-edge input z_e = act1(P[t_e] + Q[j] + E_e) of the message (apply_edges with the first Dense of phi split into
// node-level terms, src/layers.jl:523) is formed while the rows are staged instead of being read back from an [E][k] array that a
// separate launch wrote; z_out (nullable) keeps it for the pullback.
struct GnoFuse {
  const float *P, *Q, *E;      // [N][k] at the target, [N][k] at the source, [E][k] in p order; each nullable
  const int *col_s;            // target node of every entry of the by-source list
  float *z_out;                // [E][k] p order, nullable
  int act1;
};
template <int KD, bool FUSED>   // kdim (multiple of 16)
__global__ __launch_bounds__(256) void gno_apply_mfma_fwd_kernel(int cout, const int *__restrict__ rowptr_s, const int *__restrict__ xpos,
                                                                 const float *__restrict__ T, const float *__restrict__ Bh,
                                                                 const float *__restrict__ z, float *__restrict__ m, const GnoFuse fz) {
  constexpr int ZS = KD + 4;
  __shared__ __attribute__((aligned(16))) float zl[kEB * ZS];
  __shared__ int pl[kEB], tl[kEB];
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rs = rowptr_s[j], re = rowptr_s[j + 1];
  if (rs == re) return;
  const int i = lane & 15, kq = lane >> 4;
  const int nct = cout / 16;                       // output-column tiles; wave w owns tiles w, w + 4, ... (at most 4: cout <= 256)
  // B operand: T_j[o = ct*16 + i][16 kb + 4 kq .. + 3], kept for the node's whole edge list
  float4 breg[4][KD / 16];
  float bias[4];
  const float *Tj = T + (size_t)j * cout * KD;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int ct = wave + 4 * c;
    bias[c] = 0.f;
    if (ct < nct) {   // wave-uniform
#pragma unroll
      for (int kb = 0; kb < KD / 16; ++kb)
        breg[c][kb] = *reinterpret_cast<const float4 *>(Tj + (size_t)(ct * 16 + i) * KD + 16 * kb + 4 * kq);
      if (Bh) bias[c] = Bh[(size_t)j * cout + ct * 16 + i];
    }
  }
  // the pass's edge positions (and targets) are loaded a pass ahead, under the previous pass's products
  int npl = 0, ntl = 0;
  if (tid < kEB && rs + tid < re) {
    npl = xpos[rs + tid];
    if (FUSED) ntl = fz.col_s[rs + tid];
  }
  for (int q0 = rs; q0 < re; q0 += kEB) {
    const int nb = min(kEB, re - q0);
    __syncthreads();                                // the previous pass's tiles are consumed
    if (tid < kEB) {
      pl[tid] = npl;
      if (FUSED) tl[tid] = ntl;
      npl = ntl = 0;
      if (q0 + kEB + tid < re) {
        npl = xpos[q0 + kEB + tid];
        if (FUSED) ntl = fz.col_s[q0 + kEB + tid];
      }
    }
    __syncthreads();
    if constexpr (FUSED) {
      constexpr int W4 = KD / 4;
      for (int idx = tid; idx < kEB * W4; idx += 256) {
        const int e = idx / W4, c4 = idx - e * W4;
        float4 v[1] = {f4_zero()};
        if (e < nb) {
          if (fz.P) v[0] = reinterpret_cast<const float4 *>(fz.P + (size_t)tl[e] * KD)[c4];
          if (fz.Q) v[0] = f4_add(v[0], reinterpret_cast<const float4 *>(fz.Q + (size_t)j * KD)[c4]);
          if (fz.E) v[0] = f4_add(v[0], reinterpret_cast<const float4 *>(fz.E + (size_t)pl[e] * KD)[c4]);
          f4n_act<1>(fz.act1, v);
          if (fz.z_out) reinterpret_cast<float4 *>(fz.z_out + (size_t)pl[e] * KD)[c4] = v[0];
        }
        *reinterpret_cast<float4 *>(&zl[e * ZS + 4 * c4]) = v[0];
      }
    } else {
      stage_edge_rows<kEB>(zl, z, pl, nb, KD, tid);
    }
    __syncthreads();
    for (int et = 0; et * kET < nb; ++et) {         // uniform
      float4 a4[KD / 16];
#pragma unroll
      for (int kb = 0; kb < KD / 16; ++kb) a4[kb] = *reinterpret_cast<const float4 *>(&zl[(et * kET + i) * ZS + 16 * kb + 4 * kq]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ct = wave + 4 * c;
        if (ct >= nct) break;                       // wave-uniform
        f32x4 acc = (f32x4){bias[c], bias[c], bias[c], bias[c]};
#pragma unroll
        for (int kb = 0; kb < KD / 16; ++kb) {
          acc = mfma16(a4[kb].x, breg[c][kb].x, acc);
          acc = mfma16(a4[kb].y, breg[c][kb].y, acc);
          acc = mfma16(a4[kb].z, breg[c][kb].z, acc);
          acc = mfma16(a4[kb].w, breg[c][kb].w, acc);
        }
        // D[edge 4 kq + r][out ct*16 + i]: 16 lanes write 64 contiguous bytes of one edge's row
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = et * kET + 4 * kq + r;
          if (e < nb) m[(size_t)pl[e] * cout + ct * 16 + i] = acc[r];
        }
      }
    }
  }
}

// ---- pullback -------------------------------------------------------------------------------------------------------------
// NODE: the message gradient is not read from an [E][out] array but formed while the rows are staged, from the gradient of the
// sum / mean aggregation over targets: dm_e = dagg[t_e] (* 1 / deg(t_e) for mean) -- a gather from a node-level array that lives
// in L2 instead of a 231 MB stream (config 5, r = 0.1) that a separate launch wrote.
// With it the launch also finishes the pullback of the message's per-edge input z = act1(P[t] + Q[j] + E) (act1 identity / relu,
// `z` = the activated value): dz leaves already multiplied by act1'(z), and dq[j] = sum of the node's dz rows -- every edge of
// the workgroup has source j -- is formed on the way (the composed path: a by-source gather launch over the [E][k] array).
struct GnoNodeGrad {
  const float *dagg;      // [N][out]
  const int *col_s;       // target node of every entry of the by-source list
  const int *rowptr_t;    // by-target row pointers (in-degrees), read for mean only
  int mean;
  int act1;               // NGPDE_ACT_IDENTITY or NGPDE_ACT_RELU
  float *dq;              // [N][k], nullable
};
template <int KD, int NOB, bool NODE>   // NOB: bound on cout / 16 (8 or 16) -- sizes the register copy of T_j's columns
__global__ __launch_bounds__(256, NOB == 8 ? 4 : 2) void gno_apply_mfma_bwd_kernel(int cout, const int *__restrict__ rowptr_s, const int *__restrict__ xpos,
                                                                 const float *__restrict__ T, const float *__restrict__ z,
                                                                 const float *__restrict__ dm, float *__restrict__ dT,
                                                                 float *__restrict__ dBh, float *__restrict__ dz, const GnoNodeGrad ng) {
  constexpr int ZS = KD + 4;
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int DS = cout + 4;
  float *zl = sh;                                   // [kEBb][KD + 4]
  float *dml = zl + kEBb * ZS;                       // [kEBb][cout + 4]
  __shared__ int pl[kEBb], tl[kEBb];
  __shared__ float invl[kEBb < 64 ? 64 : kEBb];
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rs = rowptr_s[j], re = rowptr_s[j + 1];
  const int i = lane & 15, kq = lane >> 4;
  const int nct = cout / 16;
  constexpr int NKT = KD / 16;                      // k tiles
  // dT tiles (out tile ot, k tile kt): id = ot * NKT + kt, wave w owns ids w, w + 4, ...: at most (256/16) * (128/16) / 4 = 32
  constexpr int MAXT = 8;                           // tiles per wave kept in registers per sweep (cout * KD <= 128 * 64 in one sweep)
  const int ntile = nct * NKT;
  const float *Tj = T + (size_t)j * cout * KD;
  // dz = DM x T_j: the (edge tile, k tile) ids a wave takes (wave, wave + 4, ...) all have k tile wave % NKT (NKT divides 4), so
  // the wave keeps ITS 16 columns of T_j in registers for the node's whole edge list -- bT[ob][r] = T_j[16 ob + 4 kq + r][16 kt + i],
  // 4 registers per 16 outputs -- and T_j never touches LDS (a transposed copy of it there was 34 of the workgroup's 59 KB at
  // 128 x 64: two workgroups per CU; without it five fit)
  const int ktw = wave % NKT;
  float bT[NOB][4];
  if (dz && rs < re) {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
      if (ob < nct) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bT[ob][r] = Tj[(size_t)(16 * ob + 4 * kq + r) * KD + 16 * ktw + i];
      }
  }
  float bsum = 0.f;                                 // dBh: thread tid < cout sums column tid of the dm rows
  float qsum = 0.f;                                 // dq (NODE): this lane's share of column 16 ktw + i (its rows 4 kq + r of every tile)
  for (int sweep = 0; sweep * 4 * MAXT < ntile; ++sweep) {   // one sweep over the edge list per 32 dT tiles (a single sweep at 128 x 64)
    f32x4 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the pass's edge positions and targets are loaded a pass ahead, under the previous pass's products (mean's 1 / deg hangs
    // on the target and stays in the pass: callers that can pre-scale dagg pass it as a sum)
    int npl = 0, ntl = 0;
    if (tid < kEBb && rs + tid < re) {
      npl = xpos[rs + tid];
      if (NODE) ntl = ng.col_s[rs + tid];
    }
    for (int q0 = rs; q0 < re; q0 += kEBb) {
      const int nb = min(kEBb, re - q0);
      __syncthreads();
      if (tid < kEBb) {
        pl[tid] = npl;
        if (NODE) {
          const int t = ntl;
          tl[tid] = t;
          float inv = 1.0f;
          if (ng.mean) {
            const int deg = ng.rowptr_t[t + 1] - ng.rowptr_t[t];
            inv = 1.0f / (float)max(deg, 1);
          }
          invl[tid] = inv;
        }
        npl = ntl = 0;
        if (q0 + kEBb + tid < re) {
          npl = xpos[q0 + kEBb + tid];
          if (NODE) ntl = ng.col_s[q0 + kEBb + tid];
        }
      }
      __syncthreads();
      stage_edge_rows<kEBb>(zl, z, pl, nb, KD, tid);
      if (NODE) {
        const int w4 = cout / 4;
        for (int idx = tid; idx < kEBb * w4; idx += 256) {
          const int e = idx / w4, c4 = idx - e * w4;
          float4 v = f4_zero();
          if (e < nb) {
            v = reinterpret_cast<const float4 *>(ng.dagg + (size_t)tl[e] * cout)[c4];
            if (ng.mean) v = f4_scale(invl[e], v);
          }
          *reinterpret_cast<float4 *>(&dml[e * DS + 4 * c4]) = v;
        }
      } else {
        stage_edge_rows<kEBb>(dml, dm, pl, nb, cout, tid);
      }
      __syncthreads();
      const int net = (nb + kET - 1) / kET;
      if (sweep == 0) {
        if (dBh && tid < cout)
          for (int e = 0; e < nb; ++e) bsum += dml[e * DS + tid];
        if (dz) {
          // DZ[edge][kk] = sum_o DM[edge][o] T[o][kk]: tiles (edge tile et, k tile kt), spread over the waves
          for (int id = wave; id < net * NKT; id += 4) {
            const int et = id / NKT;   // (k tile: ktw)
            f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
              if (ob < nct) {   // uniform
                const float4 a4 = *reinterpret_cast<const float4 *>(&dml[(et * kET + i) * DS + 16 * ob + 4 * kq]);
                d = mfma16(a4.x, bT[ob][0], d);
                d = mfma16(a4.y, bT[ob][1], d);
                d = mfma16(a4.z, bT[ob][2], d);
                d = mfma16(a4.w, bT[ob][3], d);
              }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int e = et * kET + 4 * kq + r;
              if (e < nb) {
                float v = d[r];
                if (NODE) {
                  if (ng.act1 == NGPDE_ACT_RELU && !(zl[e * ZS + ktw * 16 + i] > 0.f)) v = 0.f;
                  qsum += v;
                }
                dz[(size_t)pl[e] * KD + ktw * 16 + i] = v;
              }
            }
          }
        }
      }
      if (dT) {
        // dT[o][kk] += sum_e DM[e][o] Z[e][kk]: the edge index is the contraction (zero rows beyond nb add nothing).  A wave's tiles
        // (ids wave, wave + 4, ...; NKT divides 4) all have k tile ktw, so a contraction step reads its z value ONCE for the wave's
        // eight out tiles and its eight dm values feed eight independent accumulator chains (the tile-outer form read both
        // operands per product: 16 LDS reads per 8 products instead of 9, each product waiting on its own pair)
        for (int es = 0; es < net * 4; ++es) {   // 4-edge contraction steps, in edge order per accumulator as before
          const int e = 4 * es + kq;
          const float zv = zl[e * ZS + ktw * 16 + i];
          const float *drow = dml + e * DS + i;
#pragma unroll
          for (int t = 0; t < MAXT; ++t) {
            const int id = wave + 4 * (t + MAXT * sweep);
            if (id < ntile) acc[t] = mfma16(drow[(id / NKT) * 16], zv, acc[t]);   // wave-uniform
          }
        }
      }
    }
    if (dT) {
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const int id = wave + 4 * (t + MAXT * sweep);
        if (id < ntile) {
          const int ot = id / NKT, kt = id - ot * NKT;
#pragma unroll
          for (int r = 0; r < 4; ++r) dT[(size_t)j * cout * KD + (size_t)(ot * 16 + 4 * kq + r) * KD + kt * 16 + i] = acc[t][r];
        }
      }
    }
  }
  if (dBh && tid < cout) dBh[(size_t)j * cout + tid] = bsum;
  if (NODE && ng.dq) {   // the four row groups of a wave by shuffles, then the waves that share a k tile through LDS, in wave order
    qsum += __shfl_xor(qsum, 16);
    qsum += __shfl_xor(qsum, 32);
    __syncthreads();
    if (kq == 0) invl[wave * 16 + i] = qsum;       // (64 floats of the staging tables, free by now)
    __syncthreads();
    if (tid < KD) {
      const int kt = tid >> 4;
      float sq = 0.f;
      for (int w = kt; w < 4; w += NKT) sq += invl[w * 16 + (tid & 15)];
      ng.dq[(size_t)j * KD + tid] = rs < re ? sq : 0.f;
    }
  }
}

inline bool no_gno_mfma_env() {
  const char *e = std::getenv("NGPDE_NO_GNO_MFMA");
  return e && e[0] == '1';
}
inline size_t gno_mfma_bwd_lds(int cout, int kdim) { return (size_t)(kEBb * (kdim + 4) + kEBb * (cout + 4)) * sizeof(float); }

}  // namespace

bool gno_apply_mfma_supported(int cout, int kdim) {
  return !no_gno_mfma_env() && cout % 16 == 0 && cout >= 16 && cout <= 256 && (kdim == 16 || kdim == 32 || kdim == 64) &&
         gno_mfma_bwd_lds(cout, kdim) <= 150 * 1024;
}

int32_t launch_gno_apply_mfma_fwd(const ngpde_graph *g, int cout, int kdim, const float *T, const float *Bh, const float *z, float *m,
                                  hipStream_t stream) {
  if (g->n_edges == 0) return NGPDE_OK;
  const dim3 grid((unsigned)g->n_nodes), block(256);
  const GnoFuse nofuse{};
#define NGPDE_GNO_F(KK) hipLaunchKernelGGL((gno_apply_mfma_fwd_kernel<KK, false>), grid, block, 0, stream, cout, g->by_s.rowptr, g->by_s.xpos, T, Bh, z, m, nofuse)
  switch (kdim) {
    case 16: NGPDE_GNO_F(16); break;
    case 32: NGPDE_GNO_F(32); break;
    default: NGPDE_GNO_F(64); break;
  }
#undef NGPDE_GNO_F
  NGPDE_LAUNCH_CHECK("gno_apply_mfma_fwd_kernel");
  return NGPDE_OK;
}

// the same with the message's per-edge input formed in the kernel: z_e = act1(P[t_e] + Q[s_e] + E_e)
int32_t launch_gno_message_mfma_fwd(const ngpde_graph *g, int cout, int kdim, int act1, const float *P, const float *Q, const float *E,
                                    const float *T, const float *Bh, float *z_out, float *m, hipStream_t stream) {
  if (g->n_edges == 0) return NGPDE_OK;
  const dim3 grid((unsigned)g->n_nodes), block(256);
  GnoFuse fz;
  fz.P = P; fz.Q = Q; fz.E = E; fz.col_s = g->by_s.col; fz.z_out = z_out; fz.act1 = act1;
#define NGPDE_GNO_FF(KK) hipLaunchKernelGGL((gno_apply_mfma_fwd_kernel<KK, true>), grid, block, 0, stream, cout, g->by_s.rowptr, g->by_s.xpos, T, Bh, (const float *)nullptr, m, fz)
  switch (kdim) {
    case 16: NGPDE_GNO_FF(16); break;
    case 32: NGPDE_GNO_FF(32); break;
    default: NGPDE_GNO_FF(64); break;
  }
#undef NGPDE_GNO_FF
  NGPDE_LAUNCH_CHECK("gno_apply_mfma_fwd_kernel (fused input)");
  return NGPDE_OK;
}

int32_t launch_gno_apply_mfma_bwd(const ngpde_graph *g, int cout, int kdim, const float *T, const float *z, const float *dm, float *dT,
                                  float *dBh, float *dz, hipStream_t stream, const float *dagg, int mean, int act1, float *dq) {
  if (g->n_nodes == 0) return NGPDE_OK;
  const dim3 grid((unsigned)g->n_nodes), block(256);
  const size_t lds = gno_mfma_bwd_lds(cout, kdim);
  const GnoNodeGrad ng{dagg, g->by_s.col, g->by_t.rowptr, mean, act1, dq};
#define NGPDE_GNO_B2(KK, NOB, NODE)                                                                                              \
  do {                                                                                                                           \
    if (lds > 64 * 1024)   /* beyond the default dynamic-LDS limit: raise it for this kernel (cheap, idempotent) */               \
      NGPDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gno_apply_mfma_bwd_kernel<KK, NOB, NODE>),             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                               \
    hipLaunchKernelGGL((gno_apply_mfma_bwd_kernel<KK, NOB, NODE>), grid, block, lds, stream, cout, g->by_s.rowptr, g->by_s.xpos, T, z, \
                       dm, dT, dBh, dz, ng);                                                                                     \
  } while (0)
#define NGPDE_GNO_B(KK, NOB)                                                                                                     \
  do {                                                                                                                           \
    if (dagg) NGPDE_GNO_B2(KK, NOB, true);                                                                                       \
    else NGPDE_GNO_B2(KK, NOB, false);                                                                                           \
  } while (0)
  if (cout <= 128) {
    switch (kdim) {
      case 16: NGPDE_GNO_B(16, 8); break;
      case 32: NGPDE_GNO_B(32, 8); break;
      default: NGPDE_GNO_B(64, 8); break;
    }
  } else {
    switch (kdim) {
      case 16: NGPDE_GNO_B(16, 16); break;
      case 32: NGPDE_GNO_B(32, 16); break;
      default: NGPDE_GNO_B(64, 16); break;
    }
  }
#undef NGPDE_GNO_B2
#undef NGPDE_GNO_B
  NGPDE_LAUNCH_CHECK("gno_apply_mfma_bwd_kernel");
  return NGPDE_OK;
}

}  // namespace ngpde

// edge_mlp_fused.hip -- fused forward of the edge-function layers' message path
//   m_i = aggr_{e: t_e = i}  phi( [h_i; h_j; ...] )      (/root/reference/src/layers.jl:103-111, :313-326, :402-416)
// for message MLPs phi = Dense, Dense, ... up to 64 wide: one launch does
//   gather (LDS-staged distinct source rows of the tile) -> z1_e = P[t_e] + Q[s_e] + E_e -> act ->
//   the remaining Dense layers on fp32 MFMA with the weights resident in LDS -> in-tile segmented reduction,
// so no [E][h] array is written unless the caller asks for the pre-activations (training).  The reference
// materialises gather(x, t), gather(x, s), their vcat and every layer's activations over all E edges.
//
// A workgroup (8 waves) owns one 32-row tile of the locality schedule and walks the tile's edges in chunks of
// 64 (4 MFMA row tiles); per chunk and Dense layer each wave issues 32 v_mfma_f32_16x16x4_f32.  Row r of the
// tile is summed by lane group r in edge (= COO) order: no atomics, bitwise reproducible.
#include <algorithm>

#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

#define NGPDE_LAUNCH_CHECK(name)                                                         \
  do {                                                                                   \
    hipError_t _e = hipGetLastError();                                                   \
    if (_e != hipSuccess) return fail(NGPDE_ERR_HIP, "%s launch failed: %s", name, hipGetErrorString(_e)); \
  } while (0)

constexpr int kT = 512, kW = 64, kTS = kW + 4, kChunk = 64, kGroups = 32;

struct EdgeMlpK {
  const int4 *sched;
  const int2 *halo;
  const uint8_t *slots;
  int n_tiles, h1, act1, aggr;
  const float *P, *Q, *Eterm;
  int n_tail;
  int din[3], dout[3], act[3];
  const float *wt[3], *bias[3];
  float *out;
  float *save_z[4];   // [0]: z1 [E][h1]; [k]: pre-activation of tail layer k [E][dout_k]; nullable
};

__device__ __forceinline__ int xcd_tile(int b, int nb) {
  const int x = b % 8, k = b / 8;
  const int q = nb / 8, r = nb % 8;
  return x * q + min(x, r) + k;
}

__device__ __forceinline__ float4 load4_guard(const float *base, size_t row, int width, int q) {
  return (4 * q < width) ? *reinterpret_cast<const float4 *>(base + row * width + 4 * q) : f4_zero();
}

// acc = A[64][kTS] x B for this wave's row tile (w & 3) and column tiles (w >> 2) + {0, 2}   (B transposed in LDS: Bt[col][k])
__device__ __forceinline__ void mfma_chunk64(const float *ldsA, const float *ldsBt, int wave, int lane, int kblocks, int col_tiles,
                                             f32x4 (&acc)[2]) {
  const int rt = wave & 3, cg = wave >> 2;
  const int i = lane & 15, kq = lane >> 4;
  acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float *pa = ldsA + (rt * 16 + i) * kTS + 4 * kq;
  const float *pb0 = ldsBt + (cg * 16 + i) * kTS + 4 * kq;
  const float *pb1 = ldsBt + ((cg + 2) * 16 + i) * kTS + 4 * kq;
  const bool t0 = cg < col_tiles, t1 = cg + 2 < col_tiles;   // wave-uniform
  for (int kb = 0; kb < kblocks; ++kb) {
    const float4 a4 = *reinterpret_cast<const float4 *>(pa + kb * 16);
    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
    if (t0) {
      const float4 b4 = *reinterpret_cast<const float4 *>(pb0 + kb * 16);
      acc[0] = mfma16(av[0], b4.x, acc[0]); acc[0] = mfma16(av[1], b4.y, acc[0]);
      acc[0] = mfma16(av[2], b4.z, acc[0]); acc[0] = mfma16(av[3], b4.w, acc[0]);
    }
    if (t1) {
      const float4 b4 = *reinterpret_cast<const float4 *>(pb1 + kb * 16);
      acc[1] = mfma16(av[0], b4.x, acc[1]); acc[1] = mfma16(av[1], b4.y, acc[1]);
      acc[1] = mfma16(av[2], b4.z, acc[1]); acc[1] = mfma16(av[3], b4.w, acc[1]);
    }
  }
}

template <int NTAIL>
__global__ __launch_bounds__(kT) void edge_mlp_fused_fwd_kernel(const EdgeMlpK p) {
  __shared__ __attribute__((aligned(16))) float ldsQ[(kHaloCap + 1) * kW];
  __shared__ __attribute__((aligned(16))) float ldsP[kGroups * kW];
  __shared__ __attribute__((aligned(16))) float ldsA[kChunk * kTS];
  __shared__ __attribute__((aligned(16))) float ldsWt[(NTAIL > 0 ? NTAIL : 1) * kW * kTS];
  __shared__ int ldsOff[kGroups + 1], ldsRs[kGroups];
  __shared__ __attribute__((aligned(16))) unsigned ldsSlots[kGroups * 8];   // 32 slot bytes per row
  __shared__ uint8_t ldsRowOf[kGroups * kSlotWidth];                        // tile edge k -> row of the tile
  __shared__ int ldsPe[kChunk];                                             // chunk edge -> p position, -1 past the end

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = tid >> 4, q = tid & 15;           // group g <-> row g of the tile; lane q <-> features 4q..4q+3
  const int tile = xcd_tile(blockIdx.x, p.n_tiles);
  const int h1 = p.h1;

  // ---- round 1: schedule entry, slot bytes, halo entries, P row, tail weights (transposed into LDS)
  const int4 sc = p.sched[(size_t)tile * kTileRows + grp];
  const uint4 s0 = reinterpret_cast<const uint4 *>(p.slots)[((size_t)tile * kTileRows + grp) * 2];
  const uint4 s1 = reinterpret_cast<const uint4 *>(p.slots)[((size_t)tile * kTileRows + grp) * 2 + 1];
  int2 he[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) he[k] = p.halo[(size_t)tile * kHaloCap + min(grp + k * kGroups, kHaloCap - 1)];
  const int node = max(sc.x, 0);
  const float4 prow = p.P ? load4_guard(p.P, node, h1, q) : f4_zero();
  float4 wreg[NTAIL > 0 ? NTAIL : 1][2];
  float breg[NTAIL > 0 ? NTAIL : 1][2];             // bias of this lane's two output columns (MFMA D layout)
#pragma unroll
  for (int l = 0; l < NTAIL; ++l) {
    const int j = tid % kW, kg0 = tid / kW;   // output column j, k-groups kg0 and kg0 + 8
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int k = 4 * (kg0 + 8 * ps);
      float t[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        t[r] = (k + r < p.din[l] && j < p.dout[l]) ? p.wt[l][(size_t)(k + r) * p.dout[l] + j] : 0.f;
      wreg[l][ps] = make_float4(t[0], t[1], t[2], t[3]);
      const int col = ((wave >> 2) + 2 * ps) * 16 + (lane & 15);
      breg[l][ps] = (p.bias[l] && col < p.dout[l]) ? p.bias[l][col] : 0.f;
    }
  }
  // ---- round 2: the tile's distinct source rows of Q
  float4 hv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) hv[k] = p.Q ? load4_guard(p.Q, he[k].x, h1, q) : f4_zero();

  // stage
  float4 *Q4 = reinterpret_cast<float4 *>(ldsQ);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int hh = grp + k * kGroups;
    if (hh < kHaloCap) Q4[hh * 16 + q] = hv[k];
  }
  if (grp == 0) Q4[kHaloCap * 16 + q] = f4_zero();
  reinterpret_cast<float4 *>(ldsP)[grp * 16 + q] = prow;
  if (q == 0) {
    ldsOff[grp + 1] = sc.x >= 0 ? sc.z : 0;   // degrees; turned into offsets below
    ldsRs[grp] = sc.y;
    if (grp == 0) ldsOff[0] = 0;
  }
  if (q < 8) {
    const unsigned w[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    unsigned v = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) v = (q == j) ? w[j] : v;
    ldsSlots[grp * 8 + q] = v;
  }
#pragma unroll
  for (int l = 0; l < NTAIL; ++l) {
    const int j = tid % kW, kg0 = tid / kW;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
      *reinterpret_cast<float4 *>(&ldsWt[l * kW * kTS + j * kTS + 4 * (kg0 + 8 * ps)]) = wreg[l][ps];
  }
  __syncthreads();
  if (tid == 0) {   // 32-entry prefix sum of the degrees
    int run = 0;
    for (int r = 0; r < kGroups; ++r) {
      const int d = ldsOff[r + 1];
      ldsOff[r + 1] = run + d;
      run += d;
    }
  }
  __syncthreads();
  const int total = ldsOff[kGroups];
  const int my_lo = ldsOff[grp], my_hi = ldsOff[grp + 1];
  const int last_w = (NTAIL > 0) ? p.dout[NTAIL - 1] : h1;
  for (int k = my_lo + q; k < my_hi; k += 16) ldsRowOf[k] = (uint8_t)grp;   // deg <= kSlotWidth: total <= 1024

  float4 racc;
  if (p.aggr == NGPDE_AGGR_MAX) racc = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  else if (p.aggr == NGPDE_AGGR_MIN) racc = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
  else racc = f4_zero();
  __syncthreads();

  for (int c0 = 0; c0 < total; c0 += kChunk) {
    // ---- a1 = act1(P[t] + Q[s] + E) for the chunk's edges: 2 edges per lane group
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int el = 2 * grp + u;
      const int k = c0 + el;
      float4 a = f4_zero();
      int pe_i = -1;
      if (k < total) {
        const int r = ldsRowOf[k];
        const int j = k - ldsOff[r];
        const int slot = (ldsSlots[r * 8 + (j >> 2)] >> (8 * (j & 3))) & 0xff;
        pe_i = ldsRs[r] + j;                              // position of the edge in p order
        const size_t pe = (size_t)pe_i;
        float4 z = f4_add(reinterpret_cast<const float4 *>(ldsP)[r * 16 + q], Q4[slot * 16 + q]);
        if (p.Eterm) z = f4_add(z, load4_guard(p.Eterm, pe, h1, q));
        if (p.save_z[0] && 4 * q < h1) *reinterpret_cast<float4 *>(p.save_z[0] + pe * h1 + 4 * q) = z;
        a = (4 * q < h1) ? f4_act(p.act1, z) : f4_zero();
      }
      *reinterpret_cast<float4 *>(&ldsA[el * kTS + 4 * q]) = a;
      if (q == 0) ldsPe[el] = pe_i;
    }
    __syncthreads();
    // ---- remaining Dense layers on MFMA, weights resident in LDS; bias + activation applied on the accumulators
    // and written back over the chunk's activations (no second staging tile)
#pragma unroll
    for (int l = 0; l < NTAIL; ++l) {
      f32x4 acc[2];
      const int dw = p.dout[l], col_tiles = (dw + 15) / 16;
      mfma_chunk64(ldsA, ldsWt + l * kW * kTS, wave, lane, (p.din[l] + 15) / 16, col_tiles, acc);
      __syncthreads();                                   // every wave is done reading the layer's input
      const int rt = wave & 3, cg = wave >> 2, i = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int col = (cg + 2 * m) * 16 + i;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = rt * 16 + 4 * kq + reg;
          const int pe_i = ldsPe[row];
          float v = 0.f;
          if (pe_i >= 0 && col < dw) {
            const float z = acc[m][reg] + breg[l][m];
            if (p.save_z[l + 1]) p.save_z[l + 1][(size_t)pe_i * dw + col] = z;
            v = act_apply(p.act[l], z);
          }
          ldsA[row * kTS + col] = v;
        }
      }
      __syncthreads();
    }
    // ---- segmented reduction: lane group g sums the messages of row g that fall into this chunk, in edge order
    {
      const int lo = max(my_lo, c0), hi = min(my_hi, c0 + kChunk);
      for (int k = lo; k < hi; ++k) {
        const float4 m = *reinterpret_cast<const float4 *>(&ldsA[(k - c0) * kTS + 4 * q]);
        if (p.aggr == NGPDE_AGGR_MAX) racc = make_float4(fmaxf(racc.x, m.x), fmaxf(racc.y, m.y), fmaxf(racc.z, m.z), fmaxf(racc.w, m.w));
        else if (p.aggr == NGPDE_AGGR_MIN) racc = make_float4(fminf(racc.x, m.x), fminf(racc.y, m.y), fminf(racc.z, m.z), fminf(racc.w, m.w));
        else racc = f4_add(racc, m);
      }
    }
    __syncthreads();
  }
  if (sc.x >= 0 && 4 * q < last_w) {
    const int deg = my_hi - my_lo;
    if (p.aggr == NGPDE_AGGR_MEAN) racc = deg > 0 ? f4_scale(1.0f / (float)deg, racc) : f4_zero();
    *reinterpret_cast<float4 *>(p.out + (size_t)sc.x * last_w + 4 * q) = racc;
  }
}

__global__ void activation_fwd_kernel(int64_t count, int act, const float *__restrict__ z, float *__restrict__ a) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    a[i] = act_apply(act, z[i]);
}

}  // namespace

bool edge_mlp_fused_supported(const ngpde_graph *g, const EdgeMlpArgs &a) {
  if (!g || !g->has_norm || !g->by_t.halo_ok) return false;   // needs the tile schedule, halo lists and slot bytes
  if (a.h1 <= 0 || a.h1 > kW || a.h1 % 4) return false;
  if (a.n_tail < 0 || a.n_tail > 3) return false;
  int prev = a.h1;
  for (int l = 0; l < a.n_tail; ++l) {
    if (a.din[l] != prev || a.dout[l] <= 0 || a.dout[l] > kW || a.dout[l] % 4) return false;
    prev = a.dout[l];
  }
  return true;
}

int32_t launch_edge_mlp_fused_fwd(const ngpde_graph *g, const EdgeMlpArgs &a, hipStream_t stream) {
  NGPDE_REQUIRE(edge_mlp_fused_supported(g, a), NGPDE_ERR_UNSUPPORTED,
                "fused edge-MLP path needs widths <= 64 and multiples of 4, <= 3 layers after the first, and a graph whose "
                "tiles fit the LDS halo (degree <= %d, <= %d distinct sources per 32-row tile)", kSlotWidth, kHaloCap);
  if (g->n_nodes == 0) return NGPDE_OK;
  EdgeMlpK k;
  k.sched = g->by_t.sched; k.halo = g->by_t.halo; k.slots = g->by_t.slots;
  k.n_tiles = (int)(g->n_sched / kTileRows); k.h1 = a.h1; k.act1 = a.act1; k.aggr = a.aggr;
  k.P = a.P; k.Q = a.Q; k.Eterm = a.Eterm; k.n_tail = a.n_tail;
  for (int l = 0; l < 3; ++l) {
    k.din[l] = a.din[l]; k.dout[l] = a.dout[l]; k.act[l] = a.act[l]; k.wt[l] = a.wt[l]; k.bias[l] = a.bias[l];
  }
  k.out = a.out;
  for (int l = 0; l < 4; ++l) k.save_z[l] = a.save_z[l];
  const dim3 grid(k.n_tiles), block(kT);
  switch (a.n_tail) {
    case 0: hipLaunchKernelGGL(edge_mlp_fused_fwd_kernel<0>, grid, block, 0, stream, k); break;
    case 1: hipLaunchKernelGGL(edge_mlp_fused_fwd_kernel<1>, grid, block, 0, stream, k); break;
    case 2: hipLaunchKernelGGL(edge_mlp_fused_fwd_kernel<2>, grid, block, 0, stream, k); break;
    default: hipLaunchKernelGGL(edge_mlp_fused_fwd_kernel<3>, grid, block, 0, stream, k); break;
  }
  NGPDE_LAUNCH_CHECK("edge_mlp_fused_fwd_kernel");
  return NGPDE_OK;
}

int32_t launch_activation_fwd(int64_t count, int act, const float *z, float *a, hipStream_t stream) {
  if (count == 0) return NGPDE_OK;
  const int blocks = (int)std::min<int64_t>((count + 255) / 256, 4096);
  hipLaunchKernelGGL(activation_fwd_kernel, dim3(blocks), dim3(256), 0, stream, count, act, z, a);
  NGPDE_LAUNCH_CHECK("activation_fwd_kernel");
  return NGPDE_OK;
}

}  // namespace ngpde

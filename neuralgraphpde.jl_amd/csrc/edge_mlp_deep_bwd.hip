// edge_mlp_deep_bwd.hip -- fused pullback of the edge-function layers' message path for message MLPs of THREE or FOUR Dense layers
//   m_i = aggr_{e: t_e = i} phi([h_i; h_j; ...]),  phi = Dense -> Dense -> Dense [-> Dense]
// (/root/reference/src/layers.jl:103-111, :313-326, :402-416; the VMH tutorial's phi = 4 => 60 => 60 => 60 => 40, tanh:
// /root/reference/docs/src/tutorials/VMH.md:75-83).  edge_mlp_fused.hip's pullback stops at two layers; deeper phi used to take the
// primitives' path: every layer's [E][h] pre-activations written by the forward and read back, ~12 launches.  Here, as there:
//   * the forward saves NOTHING per edge: a wave recomputes the chain z1 -> a1 -> z2 -> ... for its 16 edges in registers
//     (transposed products z^T = W^T a^T, whose result layout is again the operand layout: edge_mlp_fused.hip), keeping every
//     layer's activation and derivative;
//   * then walks back: dz_l = g . act_l'(z_l), db_l += dz_l, dW_l += a_{l-1}^T dz_l on the matrix pipe (operands transposed through
//     the wave's 4 KB of LDS), g <- W_l dz_l^T;
//   * dz1 is written once ([E][h1]: the gradient of the per-edge first-layer term and the input of the by-source sum) and summed
//     per target through LDS (dP).
// Registers are what this needs (three layers: 7 x 16 for activations / derivatives, 3 x 64 weight-gradient accumulators), so a
// workgroup is 4 waves with one wave per SIMD (512 registers per lane, accumulators in the AccVGPR half) and one workgroup per CU;
// both orientations of every tail weight stay in LDS (104 KB at three tail layers).  Sized for the tutorials' graphs (thousands of
// nodes: a launch is a handful of tiles per workgroup), not for BASELINE config 4, whose two-layer phi has its own kernels.
#include <algorithm>

#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

constexpr int kT4 = 256, kW = 64, kTS = kW + 4, kChunk4 = 64, kRows = 32, kMaxTail = 3;

struct DeepBwdK {
  const int4 *sched;
  const int2 *halo;
  const uint8_t *slots;
  int n_tiles, h1, act1, aggr, halo_rows, n_tail;
  int dout[kMaxTail], act[kMaxTail];
  const float *P, *Q, *Eterm, *dout_grad;
  const float *wt[kMaxTail], *bias[kMaxTail];
  float *dP, *dE;
  float *partial[kMaxTail];   // per tail layer: [n_workgroups][(din_l + 1)][dout_l]  (row din_l = bias gradient)
};

struct MetaD {
  int4 sc0, sc1;       // schedule rows g16 and g16 + 16
  unsigned sw0, sw1;   // slot word (q & 7) of those rows
  int he[6];           // node ids of halo rows g16 + 16 k
};

template <int NT>
__global__ __launch_bounds__(kT4, 1) void edge_mlp_deep_bwd_kernel(const DeepBwdK p) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  float *ldsQ = dyn;                                              // [halo_rows + 1][kTS]
  float *ldsP = ldsQ + (size_t)(p.halo_rows + 1) * kTS;           // [32][kTS]
  float *ldsS = ldsP + kRows * kTS;                               // [64][kTS]  wave-private transposes, then dz1 of the chunk
  float *ldsWf = ldsS + kChunk4 * kTS;                            // [NT][64 out][kTS]  W_l^T
  float *ldsWb = ldsWf + (size_t)NT * kW * kTS;                   // [NT][64 in][kTS]   W_l
  __shared__ int ldsOff[kRows + 1], ldsRs[kRows], ldsNode[kRows];
  __shared__ float ldsInv[kRows];
  __shared__ __attribute__((aligned(16))) unsigned ldsSlots[kRows * 8];
  __shared__ uint16_t ldsEdge[kRows * kSlotWidth];
  __shared__ __attribute__((aligned(16))) float ldsBias[NT * kW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g16 = tid >> 4, q = tid & 15;
  const int ei = lane & 15, kq = lane >> 4;
  const int h1 = p.h1, zero_slot = p.halo_rows;

  const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
  const int range_len = p.n_tiles / 8 + (xcd < p.n_tiles % 8 ? 1 : 0);
  const int range_lo = xcd * (p.n_tiles / 8) + min(xcd, p.n_tiles % 8);

  auto fetch_meta = [&](int tile, MetaD &m) {
    const size_t row = (size_t)tile * kTileRows + g16;
    m.sc0 = p.sched[row];
    m.sc1 = p.sched[row + 16];
    m.sw0 = reinterpret_cast<const unsigned *>(p.slots)[row * 8 + (q & 7)];
    m.sw1 = reinterpret_cast<const unsigned *>(p.slots)[(row + 16) * 8 + (q & 7)];
#pragma unroll
    for (int k = 0; k < 6; ++k) m.he[k] = p.halo[(size_t)tile * kHaloCap + min(g16 + 16 * k, kHaloCap - 1)].x;
  };

  // ---- once per workgroup: both orientations of every tail weight (zero-padded to 64 x 64), the biases, the all-zero halo row
  {
    const int j = tid & 63, kg0 = tid >> 6;
#pragma unroll
    for (int l = 0; l < NT; ++l) {
      const int din = l == 0 ? h1 : p.dout[l - 1], dw = p.dout[l];
      const float *w = p.wt[l];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int k = 4 * (kg0 + 4 * ps);
        float t[4], u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t[r] = (k + r < din && j < dw) ? w[(size_t)(k + r) * dw + j] : 0.f;        // W^T row j (output), inputs k..k+3
          u[r] = (j < din && k + r < dw) ? w[(size_t)j * dw + k + r] : 0.f;          // W row j (input), outputs k..k+3
        }
        *reinterpret_cast<float4 *>(&ldsWf[((size_t)l * kW + j) * kTS + k]) = make_float4(t[0], t[1], t[2], t[3]);
        *reinterpret_cast<float4 *>(&ldsWb[((size_t)l * kW + j) * kTS + k]) = make_float4(u[0], u[1], u[2], u[3]);
      }
      if (tid < kW) ldsBias[l * kW + tid] = (p.bias[l] && tid < dw) ? p.bias[l][tid] : 0.f;
    }
    if (g16 == 0) *reinterpret_cast<float4 *>(&ldsQ[zero_slot * kTS + 4 * q]) = f4_zero();
  }

  // weight / bias gradient accumulators of this wave, per tail layer: tile (ct, mt) <-> inputs 16 ct .. + 15 x outputs 16 mt .. + 15
  f32x4 accW[NT][4][4];
  float4 dbacc[NT][4];
#pragma unroll
  for (int l = 0; l < NT; ++l)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      dbacc[l][a] = f4_zero();
#pragma unroll
      for (int b = 0; b < 4; ++b) accW[l][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  MetaD meta;
  int jt = wg_in_xcd;
  if (jt < range_len) fetch_meta(range_lo + jt, meta);

  for (; jt < range_len; jt += wgs_per_xcd) {
    const int4 sc0 = meta.sc0, sc1 = meta.sc1;
    {   // stage the tile: P rows and the distinct Q rows
      const float4 p0 = (p.P && 4 * q < h1) ? *reinterpret_cast<const float4 *>(p.P + (size_t)max(sc0.x, 0) * h1 + 4 * q) : f4_zero();
      const float4 p1 = (p.P && 4 * q < h1) ? *reinterpret_cast<const float4 *>(p.P + (size_t)max(sc1.x, 0) * h1 + 4 * q) : f4_zero();
      float4 hv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k)
        hv[k] = (p.Q && g16 + 16 * k < p.halo_rows && 4 * q < h1) ? *reinterpret_cast<const float4 *>(p.Q + (size_t)meta.he[k] * h1 + 4 * q) : f4_zero();
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int hh = g16 + 16 * k;
        if (hh < p.halo_rows) *reinterpret_cast<float4 *>(&ldsQ[hh * kTS + 4 * q]) = hv[k];
      }
      *reinterpret_cast<float4 *>(&ldsP[g16 * kTS + 4 * q]) = p0;
      *reinterpret_cast<float4 *>(&ldsP[(g16 + 16) * kTS + 4 * q]) = p1;
    }
    if (q == 0) {
      const int d0 = sc0.x >= 0 ? sc0.z : 0, d1 = sc1.x >= 0 ? sc1.z : 0;
      ldsOff[g16 + 1] = d0;
      ldsOff[g16 + 17] = d1;
      ldsRs[g16] = sc0.y;
      ldsRs[g16 + 16] = sc1.y;
      ldsNode[g16] = max(sc0.x, 0);
      ldsNode[g16 + 16] = max(sc1.x, 0);
      ldsInv[g16] = p.aggr == NGPDE_AGGR_MEAN ? (d0 > 0 ? 1.0f / (float)d0 : 0.f) : 1.0f;
      ldsInv[g16 + 16] = p.aggr == NGPDE_AGGR_MEAN ? (d1 > 0 ? 1.0f / (float)d1 : 0.f) : 1.0f;
      if (g16 == 0) ldsOff[0] = 0;
    }
    if (q < 8) {
      ldsSlots[g16 * 8 + q] = meta.sw0;
      ldsSlots[(g16 + 16) * 8 + q] = meta.sw1;
    }
    const int jn = jt + wgs_per_xcd;
    if (jn < range_len) fetch_meta(range_lo + jn, meta);
    __syncthreads();
    if (tid < kRows) {
      int v = ldsOff[tid + 1];
#pragma unroll
      for (int o = 1; o < kRows; o <<= 1) {
        const int u = __shfl_up(v, o);
        if (tid >= o) v += u;
      }
      ldsOff[tid + 1] = v;
    }
    __syncthreads();
    const int total = ldsOff[kRows];
    const int lo0 = ldsOff[g16], hi0 = ldsOff[g16 + 1], lo1 = ldsOff[g16 + 16], hi1 = ldsOff[g16 + 17];
    for (int k = lo0 + q; k < hi0; k += 16) {
      const int j = k - lo0;
      ldsEdge[k] = (uint16_t)(g16 | (((ldsSlots[g16 * 8 + (j >> 2)] >> (8 * (j & 3))) & 0xff) << 8));
    }
    for (int k = lo1 + q; k < hi1; k += 16) {
      const int j = k - lo1;
      ldsEdge[k] = (uint16_t)((g16 + 16) | (((ldsSlots[(g16 + 16) * 8 + (j >> 2)] >> (8 * (j & 3))) & 0xff) << 8));
    }
    float4 racc0 = f4_zero(), racc1 = f4_zero();
    __syncthreads();

    for (int c0 = 0; c0 < total; c0 += kChunk4) {
      const bool wave_on = c0 + wave * 16 < total;   // wave-uniform
      const int k = c0 + wave * 16 + ei;
      const bool valid = k < total;
      float *mine = ldsS + (size_t)(wave * 16) * kTS;          // this wave's 16 rows of the staging tile
      float4 dz1[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
      if (wave_on) {
        const unsigned ew = ldsEdge[valid ? k : 0];
        const int r = ew & 0xff, slot = valid ? (int)(ew >> 8) : zero_slot;
        const size_t pe = (size_t)(ldsRs[r] + (k - ldsOff[r]));
        const int dlast = p.dout[NT - 1];
        // incoming gradient rows of the edge's target (x 1 / deg for mean): issued now, used behind the recomputed chain
        float4 g[4];
        {
          const float *grow = p.dout_grad + (size_t)ldsNode[r] * dlast + 4 * kq;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) g[mt] = (16 * mt + 4 * kq < dlast) ? *reinterpret_cast<const float4 *>(grow + 16 * mt) : f4_zero();
        }
        const float inv = valid ? ldsInv[r] : 0.f;
        // ---- the chain, recomputed: a[l] = input of tail layer l, d[l] = derivative of the activation that produced it
        // (d[0]: act1'(z1); d[l + 1]: act_l'(z_{l + 1})); padded features and edges beyond the tile are zero throughout
        float4 a[NT][4], d[NT + 1][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int f = 16 * ct + 4 * kq;
          float4 z = f4_add(*reinterpret_cast<const float4 *>(&ldsP[r * kTS + f]), *reinterpret_cast<const float4 *>(&ldsQ[slot * kTS + f]));
          if (p.Eterm && valid && f < h1) z = f4_add(z, *reinterpret_cast<const float4 *>(p.Eterm + pe * h1 + f));
          a[0][ct] = z;
          d[0][ct] = z;
        }
        f4n_act<4>(p.act1, a[0]);
        f4n_dact<4>(p.act1, d[0]);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          if (!(valid && 16 * ct + 4 * kq < h1)) a[0][ct] = f4_zero();
#pragma unroll
        for (int l = 0; l < NT; ++l) {
          const int din = l == 0 ? h1 : p.dout[l - 1], dw = p.dout[l];
          const int n_ct = (din + 15) >> 4, n_mt = (dw + 15) >> 4;   // uniform
          float4 z[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            if (mt < n_mt) {
              f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
              const float *wl = ldsWf + ((size_t)l * kW + mt * 16 + ei) * kTS + 4 * kq;
#pragma unroll
              for (int ct = 0; ct < 4; ++ct) {
                if (ct < n_ct) {
                  const float4 w4 = *reinterpret_cast<const float4 *>(wl + 16 * ct);
                  acc = mfma16(w4.x, a[l][ct].x, acc);
                  acc = mfma16(w4.y, a[l][ct].y, acc);
                  acc = mfma16(w4.z, a[l][ct].z, acc);
                  acc = mfma16(w4.w, a[l][ct].w, acc);
                }
              }
              const float4 b4 = *reinterpret_cast<const float4 *>(&ldsBias[l * kW + 16 * mt + 4 * kq]);
              z[mt] = make_float4(acc[0] + b4.x, acc[1] + b4.y, acc[2] + b4.z, acc[3] + b4.w);
            }
          }
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) d[l + 1][mt] = z[mt];
          f4n_dact<4>(p.act[l], d[l + 1]);
          if (l + 1 < NT) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a[l + 1][mt] = z[mt];
            f4n_act<4>(p.act[l], a[l + 1]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
              if (!(valid && 16 * mt + 4 * kq < dw)) a[l + 1][mt] = f4_zero();
          }
        }
        // ---- back through the tail layers
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) g[mt] = (valid && 16 * mt + 4 * kq < dlast) ? f4_scale(inv, g[mt]) : f4_zero();
#pragma unroll
        for (int l = NT - 1; l >= 0; --l) {
          const int din = l == 0 ? h1 : p.dout[l - 1], dw = p.dout[l];
          const int n_ct = (din + 15) >> 4, n_mt = (dw + 15) >> 4;   // uniform
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            g[mt] = f4_mul(g[mt], d[l + 1][mt]);                 // dz_{l+1}; zero for invalid edges / padded features
            dbacc[l][mt] = f4_add(dbacc[l][mt], g[mt]);
          }
          // dW_l += a_l^T dz over this wave's 16 edges: both operands transposed through the wave's LDS rows
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<float4 *>(&mine[ei * kTS + 16 * ct + 4 * kq]) = a[l][ct];
          float aT[4][4];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) aT[ct][sI] = mine[(4 * sI + kq) * kTS + 16 * ct + ei];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<float4 *>(&mine[ei * kTS + 16 * mt + 4 * kq]) = g[mt];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            if (mt < n_mt) {
              float dzT[4];
#pragma unroll
              for (int sI = 0; sI < 4; ++sI) dzT[sI] = mine[(4 * sI + kq) * kTS + 16 * mt + ei];
#pragma unroll
              for (int ct = 0; ct < 4; ++ct) {
                if (ct < n_ct) {
#pragma unroll
                  for (int sI = 0; sI < 4; ++sI) accW[l][ct][mt] = mfma16(aT[ct][sI], dzT[sI], accW[l][ct][mt]);
                }
              }
            }
          }
          // g <- W_l dz^T (transposed product): the gradient w.r.t. a_l
          float4 gn[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            if (ct < n_ct) {
              f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
              const float *wl = ldsWb + ((size_t)l * kW + ct * 16 + ei) * kTS + 4 * kq;
#pragma unroll
              for (int mt = 0; mt < 4; ++mt) {
                if (mt < n_mt) {
                  const float4 w4 = *reinterpret_cast<const float4 *>(wl + 16 * mt);
                  acc = mfma16(w4.x, g[mt].x, acc);
                  acc = mfma16(w4.y, g[mt].y, acc);
                  acc = mfma16(w4.z, g[mt].z, acc);
                  acc = mfma16(w4.w, g[mt].w, acc);
                }
              }
              gn[ct] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            }
          }
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) g[ct] = gn[ct];
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int f = 16 * ct + 4 * kq;
          dz1[ct] = (valid && f < h1) ? f4_mul(g[ct], d[0][ct]) : f4_zero();
          if (valid && f < h1 && p.dE) *reinterpret_cast<float4 *>(p.dE + pe * h1 + f) = dz1[ct];
        }
      }
      // ---- dz1 of the chunk -> LDS, lane group g16 sums the rows of targets g16 and g16 + 16 in edge order (= dP)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<float4 *>(&mine[ei * kTS + 16 * ct + 4 * kq]) = dz1[ct];
      __syncthreads();
      {
        const float *base = ldsS + 4 * q - c0 * kTS;
        for (int kk = max(lo0, c0); kk < min(hi0, c0 + kChunk4); ++kk) racc0 = f4_add(racc0, *reinterpret_cast<const float4 *>(base + kk * kTS));
        for (int kk = max(lo1, c0); kk < min(hi1, c0 + kChunk4); ++kk) racc1 = f4_add(racc1, *reinterpret_cast<const float4 *>(base + kk * kTS));
      }
      __syncthreads();
    }
    if (p.dP && 4 * q < h1) {
      if (sc0.x >= 0) *reinterpret_cast<float4 *>(p.dP + (size_t)sc0.x * h1 + 4 * q) = racc0;
      if (sc1.x >= 0) *reinterpret_cast<float4 *>(p.dP + (size_t)sc1.x * h1 + 4 * q) = racc1;
    }
  }

  // ---- per layer: fold the waves' accumulators into this workgroup's slab, wave by wave (fixed order), and write it out
#pragma unroll
  for (int l = 0; l < NT; ++l) {
    const int din = l == 0 ? h1 : p.dout[l - 1], dw = p.dout[l];
    float *slab = ldsS;                                            // [(din + 1)][dw] <= 65 x 64 floats <= [64][kTS]
    __syncthreads();
    for (int idx = tid; idx < (din + 1) * dw; idx += kT4) slab[idx] = 0.f;
    float4 dbl[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {   // db: sum the 16 edge lanes of each k-quarter inside the wave first
      float v[4] = {dbacc[l][mt].x, dbacc[l][mt].y, dbacc[l][mt].z, dbacc[l][mt].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) v[c] += __shfl_xor(v[c], o);
      }
      dbl[mt] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    for (int w = 0; w < kT4 / 64; ++w) {
      if (wave == w) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kin = 16 * ct + 4 * kq + r, o = 16 * mt + ei;
              if (kin < din && o < dw) slab[kin * dw + o] += accW[l][ct][mt][r];
            }
        if (ei == 0) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int o = 16 * mt + 4 * kq;
            if (o < dw) {
              slab[din * dw + o] += dbl[mt].x; slab[din * dw + o + 1] += dbl[mt].y;
              slab[din * dw + o + 2] += dbl[mt].z; slab[din * dw + o + 3] += dbl[mt].w;
            }
          }
        }
      }
      __syncthreads();
    }
    float *dst = p.partial[l] + (size_t)blockIdx.x * (din + 1) * dw;
    for (int idx = tid; idx < (din + 1) * dw; idx += kT4) dst[idx] = slab[idx];
  }
}

int deep_grid(const ngpde_graph *g) {
  const int n_tiles = (int)(g->n_sched / kTileRows);
  return 8 * std::max(1, std::min(32, (n_tiles + 7) / 8));   // one persistent workgroup per CU, a multiple of the 8 XCDs
}

}  // namespace

bool edge_mlp_deep_bwd_supported(const ngpde_graph *g, int h1, int n_tail, const int *dout, int aggr) {
  if (!g || !g->has_norm || !g->by_t.halo_ok) return false;
  if (h1 <= 0 || h1 > kW || h1 % 4 || n_tail < 2 || n_tail > kMaxTail || !dout) return false;
  for (int l = 0; l < n_tail; ++l)
    if (dout[l] <= 0 || dout[l] > kW || dout[l] % 4) return false;
  return aggr == NGPDE_AGGR_SUM || aggr == NGPDE_AGGR_MEAN;
}

size_t edge_mlp_deep_bwd_workspace(const ngpde_graph *g, int h1, int n_tail, const int *dout) {
  size_t bytes = 256;
  int din = h1;
  for (int l = 0; l < n_tail; ++l) {
    bytes += ((size_t)deep_grid(g) * (din + 1) * dout[l] * sizeof(float) + 255) / 256 * 256;
    din = dout[l];
  }
  return bytes;
}

int32_t launch_edge_mlp_deep_bwd(const ngpde_graph *g, const EdgeMlpDeepBwdArgs &a, hipStream_t stream) {
  NGPDE_REQUIRE(edge_mlp_deep_bwd_supported(g, a.h1, a.n_tail, a.dout, a.aggr), NGPDE_ERR_UNSUPPORTED,
                "deep fused edge-MLP pullback needs widths <= 64 and multiples of 4, 2 or 3 layers after the first, + or mean "
                "aggregation and a graph whose tiles fit the LDS halo");
  if (g->n_nodes == 0) return NGPDE_OK;
  const size_t need = edge_mlp_deep_bwd_workspace(g, a.h1, a.n_tail, a.dout);
  NGPDE_REQUIRE(a.workspace && a.workspace_bytes >= need, NGPDE_ERR_WORKSPACE, "deep fused edge-MLP pullback: workspace too small (%zu < %zu bytes)",
                a.workspace_bytes, need);
  NGPDE_REQUIRE(a.dE != nullptr || g->n_edges == 0, NGPDE_ERR_INVALID_ARGUMENT, "deep fused edge-MLP pullback: the [E][h1] buffer dE is required");
  DeepBwdK k;
  k.sched = g->by_t.sched; k.halo = g->by_t.halo; k.slots = g->by_t.slots;
  k.n_tiles = (int)(g->n_sched / kTileRows); k.h1 = a.h1; k.act1 = a.act1; k.aggr = a.aggr; k.n_tail = a.n_tail;
  k.halo_rows = std::max<int>(kTileRows, std::min<int>(kHaloCap, g->by_t.max_halo));
  k.P = a.P; k.Q = a.Q; k.Eterm = a.Eterm; k.dout_grad = a.dout_grad; k.dP = a.dP; k.dE = a.dE;
  const int grid = deep_grid(g);
  char *ws = reinterpret_cast<char *>(a.workspace);
  int din = a.h1;
  for (int l = 0; l < kMaxTail; ++l) {
    k.dout[l] = l < a.n_tail ? a.dout[l] : 0; k.act[l] = l < a.n_tail ? a.act[l] : 0;
    k.wt[l] = l < a.n_tail ? a.wt[l] : nullptr; k.bias[l] = l < a.n_tail ? a.bias[l] : nullptr;
    k.partial[l] = nullptr;
    if (l < a.n_tail) {
      k.partial[l] = reinterpret_cast<float *>(ws);
      ws += ((size_t)grid * (din + 1) * a.dout[l] * sizeof(float) + 255) / 256 * 256;
      din = a.dout[l];
    }
  }
  const size_t lds = ((size_t)(k.halo_rows + 1) * kTS + (size_t)kRows * kTS + (size_t)kChunk4 * kTS + 2 * (size_t)a.n_tail * kW * kTS) * sizeof(float);
  auto launch = [&](auto kernel) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kT4), lds, stream, k);
    return hipSuccess;
  };
  const hipError_t le = a.n_tail == 2 ? launch(edge_mlp_deep_bwd_kernel<2>) : launch(edge_mlp_deep_bwd_kernel<3>);
  if (le != hipSuccess) return fail(NGPDE_ERR_HIP, "edge_mlp_deep_bwd_kernel: LDS request of %zu bytes refused: %s", lds, hipGetErrorString(le));
  NGPDE_LAUNCH_CHECK("edge_mlp_deep_bwd_kernel");
  int32_t st;
  din = a.h1;
  for (int l = 0; l < a.n_tail; ++l) {
    if ((st = launch_dense_weight_reduce(grid, din, a.dout[l], k.partial[l], a.dwt[l], a.dbias[l], stream))) return st;
    din = a.dout[l];
  }
  if (a.dQ && (st = launch_edge_sum_by_source(g, a.h1, a.dE, a.dQ, stream))) return st;
  return NGPDE_OK;
}

}  // namespace ngpde

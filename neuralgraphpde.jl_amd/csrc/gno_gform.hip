// gno_gform.hip -- GNOConv's aggregated message (/root/reference/src/layers.jl:523-534) in the aggregate-then-transform form.
//
//   m_i = aggr_{e -> i} K_e h_{s_e},   K_e = reshape(W2 z_e + b2, out, in),   z_e = act1(P[i] + Q[s_e] + E_e)        (:523-530)
//
// gno_mfma.hip reassociates this by SOURCE (T_j = W2 (x) h_j, m_e = T_j z_e): the [E][out] message array is written, read back by
// the segmented sum, and every source's 32 KB T_j is streamed once per launch (config 5, r = 0.1: 1.1 GB moved for 10.5 MB of
// compulsory bytes).  The other reassociation contracts the EDGE index first, per TARGET over its CSR row:
//
//   G_i [k][i'] = sum_{e -> i} z_e[k] h_{s_e}[i']        (= Z_i^T H_i : [k x deg] x [deg x in] on the matrix pipe, this file)
//   m_i [o]     = sum_{k,i'} W2[k][o + out i'] G_i[k][i']  +  sum_{i'} b2[o + out i'] hsum_i[i'],   hsum_i = sum_{e -> i} h_{s_e}
//
// -- the second line is ONE node-level GEMM (N x k.in) x (k.in x out) against phi's last weight matrix exactly as it lies in memory
// (row k in + i', column o: no transposed copy), split over the contraction.  Same FLOPs as the by-source form (2 k in E edge level
// + 2 N k in out node level); no per-edge output, no scatter, no atomics: G (N x 32 KB) is written once and read once, h and Q rows
// come from L2.  mean: G_i and hsum_i leave the launch divided by deg(i).
// One workgroup (4 waves) per target; edges in chunks of 32: rows of h and Q go memory -> registers a chunk ahead, z is formed on the
// way into LDS (and kept in z_out [E][k], p order, for the pullback), wave w owns G_i's rows 16 w .. 16 w + 15 (all `in` columns).
// Shapes: k = 64, in in {32, 64, 128}; anything else keeps the by-source form.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "device_utils.h"
#include "persistent_mem.h"

namespace ngpde {

// dense_mfma.hip: C [M][N] = A [M][K] x B [K][N] on 128 x 128 tiles with the contraction split over blockIdx.z (split z writes slab z),
// plus up to two short products of the same output shape in the same launch (slabs nsplit, nsplit + 1)
int32_t launch_gemm128_split_nn(int M, int N, int K, int nsplit, const float *A, int lda, const float *B, int ldb, float *slabs,
                                size_t slab_stride, int ldc, int n_side, const float *const *side_a, const float *const *side_b, const int *side_k,
                                hipStream_t stream);

namespace {

constexpr int kGK = 64;    // k (width of phi's hidden layer)

// GC: edges per chunk.  The launch is bound by how many workgroups a CU holds (a target's prologue -- row pointers, sources, the first
// rows: three dependent memory round trips -- and its 32 KB epilogue run no matrix instruction): 16-edge chunks and a two-tile
// epilogue patch keep a workgroup at 14 KB of LDS, eight workgroups per CU (the wave limit) instead of five at 32 edges.
template <int CIN, int GC>
struct GGeo {
  static constexpr int H4 = CIN / 4;            // float4 per row of h
  static constexpr int RG = 256 / H4;           // rows of a chunk one pass covers = row groups of the hsum reduction
  static constexpr int HP = (GC + RG - 1) / RG; // float4 of h per thread and chunk
  static constexpr int ZP = GC / 16;            // float4 of z per thread and chunk
  static constexpr int HS = CIN + 16;           // LDS row strides: = 16 mod 32, so the four edges (kq) of an operand read land on
  static constexpr int ZS = kGK + 16;           // disjoint halves of the 32 banks (ds_read_b32: lanes 0-31 = kq 0,1; 32-63 = kq 2,3)
  static constexpr int NT = CIN / 16;           // column tiles of G_i
  static constexpr int NTGW = GC <= 16 ? 2 : 4; // column tiles per pass of the epilogue
  static constexpr int NTG = NT < NTGW ? NT : NTGW;
  static constexpr int PS = 16 * NTG + 4;       // patch row stride
  static constexpr int kTiles = GC * (HS + ZS);
  static constexpr int kEpi = 4 * 16 * PS + RG * CIN;
  static constexpr int kLds = kTiles > kEpi ? kTiles : kEpi;
};

struct GFormArgs {
  const int *rowptr_t, *col_t;
  const float *P, *Q, *Et, *h;   // [N][64] at the target, [N][64] at the source, [E][64] p order (each nullable); [N][CIN]
  float *G, *hsum, *z_out;       // [N][64][CIN]; [N][CIN] or null; [E][64] p order or null
  int act1, mean;
};

// ACT1: phi's first activation at compile time (NGPDE_ACT_IDENTITY / NGPDE_ACT_RELU; -1: p.act1 at run time).
// The staging is written for instruction count: on this chip an fp32 MFMA and a VALU instruction of the same SIMD exclude each other
// (tools/mfma_valu_overlap.hip), so every VALU slot of the staging is matrix time lost.  First form of this kernel: ~350 VALU slots
// per wave and 32-edge chunk beside 64 MFMAs -- 39 us of a 110 us launch with the products, loads and stores all switched off.  Here:
// rows beyond a partial chunk's end are CLAMPED to its last edge instead of selected to zero (a zero z row cancels them; the
// neighbour sum takes them with a 0 / 1 weight), gathers use a scalar base + 32-bit offset, full chunks skip the masks.
template <int CIN, int kGC, int ACT1>
__global__ __launch_bounds__(256, 3) void gno_gform_fwd_kernel(const GFormArgs p) {
  using GG = GGeo<CIN, kGC>;
  __shared__ __attribute__((aligned(16))) float lds[GG::kLds];
  __shared__ int cl[2][kGC];   // a chunk's source nodes, double-buffered: chunk c reads cl[c & 1]
  float *ldsH = lds, *ldsZ = lds + kGC * GG::HS;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int rs = p.rowptr_t[n], re = p.rowptr_t[n + 1];
  const int deg = re - rs;
  const float inv = p.mean ? 1.0f / (float)max(deg, 1) : 1.0f;
  // staging roles: h rows -- float4 c4h of row (tid / H4) + RG * pass; z rows -- float4 c4z of row (tid / 16) + 16 * pass
  const int c4h = tid % GG::H4, rh = tid / GG::H4, c4z = tid & 15, rz = tid >> 4;
  const unsigned hoff = (unsigned)c4h * 16u, qoff = (unsigned)c4z * 16u;
  const float4 p4 = p.P ? reinterpret_cast<const float4 *>(p.P + (size_t)n * kGK)[c4z] : f4_zero();
  float4 hreg[GG::HP], qreg[GG::ZP], ereg[GG::ZP];
  float4 hpart = f4_zero();
  int ncol = (tid < kGC && rs + tid < re) ? p.col_t[rs + tid] : 0;
  if (tid < kGC) cl[0][tid] = ncol;
  ncol = (tid < kGC && rs + kGC + tid < re) ? p.col_t[rs + kGC + tid] : 0;
  __syncthreads();
  // rows of chunk q0 into registers (sources from cs; a row beyond the chunk's nb is the chunk's LAST edge again).  Loads only: any
  // arithmetic on a loaded value here would be waited for in front of the products the loads are meant to fly under.
  auto load_chunk = [&](int q0, int nb, const int *cs) {
#pragma unroll
    for (int ps = 0; ps < GG::HP; ++ps) {
      const int r = min(rh + GG::RG * ps, nb - 1);
      hreg[ps] = ld4_g(p.h, (unsigned)cs[r] * (unsigned)(CIN * 4) + hoff);
    }
#pragma unroll
    for (int ps = 0; ps < GG::ZP; ++ps) {
      const int r = min(rz + 16 * ps, nb - 1);
      if (p.Q) qreg[ps] = ld4_g(p.Q, (unsigned)cs[r] * (unsigned)(kGK * 4) + qoff);
      if (p.Et) ereg[ps] = reinterpret_cast<const float4 *>(p.Et + (size_t)(q0 + r) * kGK)[c4z];
    }
  };
  f32x4 acc[GG::NT];
#pragma unroll
  for (int nt = 0; nt < GG::NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (deg > 0) load_chunk(rs, min(kGC, deg), cl[0]);
  int ci = 0;
  for (int q0 = rs; q0 < re; q0 += kGC, ++ci) {
    const int nb = min(kGC, re - q0);
    const bool full = nb == kGC;   // uniform
    // registers -> LDS tiles
#pragma unroll
    for (int ps = 0; ps < GG::HP; ++ps) {
      const int r = rh + GG::RG * ps;
      const float4 v = hreg[ps];
      if (p.hsum) {   // uniform
        if (full) hpart = f4_add(hpart, v);
        else hpart = f4_fma(r < nb ? 1.0f : 0.0f, v, hpart);
      }
      if (GG::HP * GG::RG == kGC || r < kGC) *reinterpret_cast<float4 *>(&ldsH[r * GG::HS + 4 * c4h]) = v;
    }
    {
      float4 zz[GG::ZP];
#pragma unroll
      for (int ps = 0; ps < GG::ZP; ++ps) {
        zz[ps] = p4;
        if (p.Q) zz[ps] = f4_add(zz[ps], qreg[ps]);
        if (p.Et) zz[ps] = f4_add(zz[ps], ereg[ps]);
      }
      if constexpr (ACT1 == NGPDE_ACT_RELU) {   // one v_med3 per value (max(x, 0) costs a canonicalising v_max in front of it)
        const float big = __builtin_inff();
#pragma unroll
        for (int ps = 0; ps < GG::ZP; ++ps)
          zz[ps] = make_float4(__builtin_amdgcn_fmed3f(zz[ps].x, 0.f, big), __builtin_amdgcn_fmed3f(zz[ps].y, 0.f, big),
                               __builtin_amdgcn_fmed3f(zz[ps].z, 0.f, big), __builtin_amdgcn_fmed3f(zz[ps].w, 0.f, big));
      } else if constexpr (ACT1 != NGPDE_ACT_IDENTITY) {
        f4n_act<GG::ZP>(p.act1, zz);
      }
#pragma unroll
      for (int ps = 0; ps < GG::ZP; ++ps) {
        const int r = rz + 16 * ps;
        const float4 v = full ? zz[ps] : f4_scale(r < nb ? 1.0f : 0.0f, zz[ps]);   // (a zero z row: the row's h does not count)
        if (p.z_out && r < nb) reinterpret_cast<float4 *>(p.z_out + (size_t)(q0 + r) * kGK)[c4z] = v;
        *reinterpret_cast<float4 *>(&ldsZ[r * GG::ZS + 4 * c4z]) = v;
      }
    }
    if (tid < kGC) cl[(ci + 1) & 1][tid] = ncol;   // the next chunk's sources (that buffer was last read two iterations ago)
    __syncthreads();
    if (q0 + kGC < re) {
      load_chunk(q0 + kGC, min(kGC, re - q0 - kGC), cl[(ci + 1) & 1]);   // in flight under the products
      ncol = (tid < kGC && q0 + 2 * kGC + tid < re) ? p.col_t[q0 + 2 * kGC + tid] : 0;
    }
    // G_i[16 wave + m][16 nt + c] += sum over the chunk's edges of z_e[16 wave + m] h_{s_e}[16 nt + c]: one k-step = four edges
    // (unrolled over the chunk's at most GC / 4 k-steps with two operand register sets taking turns: written as a loop with
    // `cur = nxt` the compiler keeps the sets apart with 18 v_mov per 8 MFMAs -- and on this chip every VALU cycle is an MFMA cycle lost)
    const int nks = (nb + 3) >> 2;
    const float *za = ldsZ + kq * GG::ZS + 16 * wave + i, *hb = ldsH + kq * GG::HS + i;
    float af[2], bf[2][GG::NT];
    af[0] = za[0];
#pragma unroll
    for (int nt = 0; nt < GG::NT; ++nt) bf[0][nt] = hb[16 * nt];
#pragma unroll
    for (int ks = 0; ks < kGC / 4; ++ks) {
      if (ks < nks) {   // uniform
        if (ks + 1 < kGC / 4 && ks + 1 < nks) {
          af[(ks + 1) & 1] = za[(ks + 1) * 4 * GG::ZS];
#pragma unroll
          for (int nt = 0; nt < GG::NT; ++nt) bf[(ks + 1) & 1][nt] = hb[(ks + 1) * 4 * GG::HS + 16 * nt];
        }
#pragma unroll
        for (int nt = 0; nt < GG::NT; ++nt) acc[nt] = mfma16(af[ks & 1], bf[ks & 1][nt], acc[nt]);
      }
    }
    __syncthreads();   // the tiles are consumed
  }
  // epilogue: this wave's 16 rows of G_i through its own LDS patch, NTG column tiles at a time, as 16-byte stores of whole row segments
  float *patch = lds + wave * (16 * GG::PS);
  float *red = lds + 4 * 16 * GG::PS;
  if (p.hsum) *reinterpret_cast<float4 *>(&red[rh * CIN + 4 * c4h]) = hpart;
  float *Gn = p.G + ((size_t)n * kGK + 16 * wave) * CIN;
  constexpr int P4 = 4 * GG::NTG;   // float4 per patch row
#pragma unroll
  for (int ps = 0; ps < GG::NT / GG::NTG; ++ps) {
#pragma unroll
    for (int j = 0; j < GG::NTG; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) patch[(4 * kq + reg) * GG::PS + 16 * j + i] = acc[GG::NTG * ps + j][reg] * inv;
#pragma unroll
    for (int r = 0; r < 16 * P4 / 64; ++r) {
      const int idx = lane + 64 * r, row = idx / P4, c4 = idx % P4;
      *reinterpret_cast<float4 *>(Gn + (size_t)row * CIN + 16 * GG::NTG * ps + 4 * c4) = *reinterpret_cast<const float4 *>(&patch[row * GG::PS + 4 * c4]);
    }
  }
  if (p.hsum) {
    __syncthreads();
    if (tid < GG::H4) {
      float4 s = f4_zero();
#pragma unroll
      for (int r = 0; r < GG::RG; ++r) s = f4_add(s, *reinterpret_cast<const float4 *>(&red[r * CIN + 4 * tid]));
      reinterpret_cast<float4 *>(p.hsum + (size_t)n * CIN)[tid] = f4_scale(inv, s);
    }
  }
}

// y = act.(sum of the slabs + bias): the layer's tail (src/layers.jl:536) fused with the slab reduction
// (slabs: the layer's workspace, 16-byte aligned; bias, y, zt are the caller's: 4-byte accesses unless `vec`)
__global__ void gno_gform_finish_kernel(int64_t count4, int cout4, int act, int nslab, size_t slab_stride4, const float4 *__restrict__ slabs,
                                        const float *__restrict__ bias, float *__restrict__ y, float *__restrict__ zt, int vec) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= count4) return;
  float4 s = slabs[idx];
  for (int z = 1; z < nslab; ++z) s = f4_add(s, slabs[(size_t)z * slab_stride4 + idx]);
  if (bias) {
    const float *b = bias + 4 * (idx % cout4);
    s = f4_add(s, make_float4(b[0], b[1], b[2], b[3]));
  }
  float4 a[1] = {s};
  f4n_act<1>(act, a);
  if (vec) {
    if (zt) reinterpret_cast<float4 *>(zt)[idx] = s;
    reinterpret_cast<float4 *>(y)[idx] = a[0];
  } else {
    if (zt) { zt[4 * idx] = s.x; zt[4 * idx + 1] = s.y; zt[4 * idx + 2] = s.z; zt[4 * idx + 3] = s.w; }
    y[4 * idx] = a[0].x; y[4 * idx + 1] = a[0].y; y[4 * idx + 2] = a[0].z; y[4 * idx + 3] = a[0].w;
  }
}

inline bool no_gform_env() {
  const char *e = std::getenv("NGPDE_NO_GNO_GFORM");
  return e && e[0] == '1';
}

}  // namespace

}  // namespace ngpde

using namespace ngpde;

extern "C" {

int32_t ngpde_gno_gform_supported(int32_t in_chs, int32_t kdim) {
  return (!no_gform_env() && kdim == kGK && (in_chs == 32 || in_chs == 64 || in_chs == 128)) ? 1 : 0;
}

int32_t ngpde_gno_gform_preferred(int64_t n_nodes, int64_t n_edges, int32_t in_chs, int32_t kdim, int32_t cout, int32_t training) {
  // Inference: always (config 5: 0.279 -> 0.200 ms at radius 0.1, 0.191 -> 0.147 ms at 0.05).  Training: the pullback stays in the
  // by-source form and needs T = W2 (x) h, which only that form's forward leaves behind -- one more node-level product (86 us at
  // config 5) that this form's forward must win back on the edges: it does from about 64 edges per node (radius 0.1, 117 per node:
  // forward + backward 0.895 -> 0.861 ms; radius 0.05, 34 per node: 0.623 -> 0.657 ms, so not there).
  if (ngpde_gno_gform_supported(in_chs, kdim) != 1 || cout % 4 != 0 || n_edges <= 0) return 0;
  if ((uint64_t)n_nodes * (uint64_t)in_chs * 4u >= (1ull << 32)) return 0;
  return (!training || n_edges >= 64 * n_nodes) ? 1 : 0;
}

int32_t ngpde_gno_gform_splits(int64_t n_nodes, int32_t in_chs, int32_t kdim, int32_t cout) {
  // enough workgroups for two per CU; every split a multiple of 16 of the contraction
  const int64_t tiles = ((n_nodes + 127) / 128) * ((cout + 127) / 128);
  const int K = in_chs * kdim;
  int ns = (int)std::max<int64_t>(1, (512 + tiles - 1) / std::max<int64_t>(tiles, 1));
  ns = std::min(ns, std::max(1, K / 256));
  return std::min(ns, 32);
}

int32_t ngpde_gno_gform_aggregate(const ngpde_graph_t *g, int32_t in_chs, int32_t kdim, int32_t act1, int32_t mean, const float *p_target,
                                  const float *q_source, const float *e_term, const float *h, float *gout, float *hsum, float *z_out,
                                  ngpde_stream_t stream) {
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_gform_aggregate: graph is NULL");
  NGPDE_REQUIRE(ngpde_gno_gform_supported(in_chs, kdim) == 1, NGPDE_ERR_UNSUPPORTED,
                "ngpde_gno_gform_aggregate: needs k = 64 and in in {32, 64, 128}, got in = %d, k = %d", in_chs, kdim);
  NGPDE_REQUIRE(act1 >= NGPDE_ACT_IDENTITY && act1 <= NGPDE_ACT_SOFTPLUS, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_gform_aggregate: unknown activation %d", act1);
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(h && gout && (p_target || q_source || e_term), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_gform_aggregate: NULL argument");
  NGPDE_REQUIRE((uint64_t)g->n_nodes * (uint64_t)in_chs * 4u < (1ull << 32), NGPDE_ERR_UNSUPPORTED,
                "ngpde_gno_gform_aggregate: h beyond 4 GB (rows are fetched with 32-bit offsets)");
  NGPDE_REQUIRE(((reinterpret_cast<uintptr_t>(p_target) | reinterpret_cast<uintptr_t>(q_source) | reinterpret_cast<uintptr_t>(e_term) | reinterpret_cast<uintptr_t>(h) |
                  reinterpret_cast<uintptr_t>(gout) | reinterpret_cast<uintptr_t>(hsum) | reinterpret_cast<uintptr_t>(z_out)) & 15) == 0,
                NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_gform_aggregate: every array must be 16-byte aligned");
  GFormArgs a;
  a.rowptr_t = g->by_t.rowptr; a.col_t = g->by_t.col;
  a.P = p_target; a.Q = q_source; a.Et = e_term; a.h = h; a.G = gout; a.hsum = hsum; a.z_out = z_out; a.act1 = act1; a.mean = mean ? 1 : 0;
  const dim3 grid((unsigned)g->n_nodes), block(256);
  hipStream_t s = (hipStream_t)stream;
  static const int chunk = [] { const char *e = std::getenv("NGPDE_GNO_GFORM_CHUNK"); return e ? std::atoi(e) : 32; }();   // (A/B runs: 16 or 32)
#define NGPDE_GF2(CC, AA)                                                                         \
  do {                                                                                            \
    if (chunk == 16) hipLaunchKernelGGL((gno_gform_fwd_kernel<CC, 16, AA>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((gno_gform_fwd_kernel<CC, 32, AA>), grid, block, 0, s, a);             \
  } while (0)
#define NGPDE_GF(CC)                                              \
  do {                                                            \
    if (act1 == NGPDE_ACT_RELU) NGPDE_GF2(CC, NGPDE_ACT_RELU);    \
    else if (act1 == NGPDE_ACT_IDENTITY) NGPDE_GF2(CC, NGPDE_ACT_IDENTITY); \
    else NGPDE_GF2(CC, -1);                                       \
  } while (0)
  switch (in_chs) {
    case 32: NGPDE_GF(32); break;
    case 64: NGPDE_GF(64); break;
    default: NGPDE_GF(128); break;
  }
#undef NGPDE_GF
#undef NGPDE_GF2
  NGPDE_LAUNCH_CHECK("gno_gform_fwd_kernel");
  return NGPDE_OK;
}

int32_t ngpde_gno_gform_transform(int64_t n_nodes, int32_t in_chs, int32_t kdim, int32_t cout, int32_t act, const float *gin, const float *w2,
                                  const float *hsum, const float *b2, const float *h, const float *w, const float *bias, float *y, float *zt,
                                  float *slabs, int32_t nsplit, ngpde_stream_t stream) {
  NGPDE_REQUIRE(n_nodes >= 0 && in_chs > 0 && kdim > 0 && cout > 0 && cout % 4 == 0 && in_chs % 16 == 0, NGPDE_ERR_DIMENSION_MISMATCH,
                "ngpde_gno_gform_transform: needs out a multiple of 4 and in a multiple of 16, got in = %d, k = %d, out = %d", in_chs, kdim, cout);
  NGPDE_REQUIRE(act >= NGPDE_ACT_IDENTITY && act <= NGPDE_ACT_SOFTPLUS, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_gform_transform: unknown activation %d", act);
  if (n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(gin && w2 && y && slabs && nsplit >= 1 && (hsum == nullptr) == (b2 == nullptr) && (h == nullptr) == (w == nullptr),
                NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_gform_transform: NULL argument (hsum / b2 and h / w come in pairs)");
  NGPDE_REQUIRE(((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(hsum) | reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(h) |
                  reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(gin) | reinterpret_cast<uintptr_t>(w2)) & 15) == 0,
                NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_gform_transform: g, w2, hsum, b2, h, w and the slabs must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int K = in_chs * kdim;
  const size_t slab = (size_t)n_nodes * cout;
  const float *sa[2], *sb[2];
  int sk[2], ns = 0;
  if (hsum) { sa[ns] = hsum; sb[ns] = b2; sk[ns] = in_chs; ++ns; }
  if (h) { sa[ns] = h; sb[ns] = w; sk[ns] = in_chs; ++ns; }
  int32_t st;
  if ((st = launch_gemm128_split_nn((int)n_nodes, cout, K, nsplit, gin, K, w2, cout, slabs, slab, cout, ns, sa, sb, sk, s))) return st;
  const int64_t count4 = (int64_t)(slab / 4);
  const int vec = ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(zt)) & 15) == 0 ? 1 : 0;
  hipLaunchKernelGGL(gno_gform_finish_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, s, count4, cout / 4, act, nsplit + ns, slab / 4,
                     reinterpret_cast<const float4 *>(slabs), bias, y, zt, vec);
  NGPDE_LAUNCH_CHECK("gno_gform_finish_kernel");
  return NGPDE_OK;
}

}  // extern "C"

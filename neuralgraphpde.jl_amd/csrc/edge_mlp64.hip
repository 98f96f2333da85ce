// edge_mlp64.hip -- the message path of the edge-function layers for the 64-wide two-layer message MLP
//   m_i = aggr_{e: t_e = i}  act2( W2^T act1( P[t_e] + Q[s_e] ) + b2 )      (/root/reference/src/layers.jl:409-416, :316-326,
//                                                                            :103-111 with the first Dense split at node level)
// in the shape BASELINE's config 4 runs (MPPDEConv, h = 64, phi = Dense(. => 64) -> Dense(64 => 64)), as a software-pipelined
// specialisation of edge_mlp_fused.hip: widths and activations are compile-time constants, so the steady state of the edge loop is
// ONE basic block in which a wave's 64 v_mfma_f32_16x16x4_f32 of slice s run beside the bias + activation + LDS store of slice
// s - 1 and the gather + first activation of slice s + 1.  The general kernel walks the same phases one after the other with every
// wave of the SIMD in the same phase (identical waves stay in step), so its matrix pipe idles during the activations and its
// VALU during the products (matrix pipe 39 % busy at config 4); and its 128-edge chunks pad a 192-edge tile to 256.
//
// A workgroup = 4 waves owns a 32-row tile of the locality schedule and walks the tile's edges 64 (4 x 16) at a time; tile
// metadata, P rows and the distinct Q rows (halo) of the NEXT tile are in flight in registers under the current tile's arithmetic
// (edge_mlp_fused.hip's scheme).  Register layout of a slice, transposed products and the in-tile segmented reduction in edge
// (= COO) order are those of edge_mlp_fused.hip: no atomics, bitwise reproducible, and bitwise equal to the general kernel.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

template <int S, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (S < N) {
    f(std::integral_constant<int, S>{});
    static_for<S + 1, N>(f);
  }
}

// Pins a value to this point of the instruction stream: the (empty) volatile asm is ordered against the other pins and the
// sched_barriers of the pipelined block, so what computes the value stays before it and what uses it after it.  Without the pins
// instruction selection places pure arithmetic next to its use -- the whole activation of the next slice behind the last MFMA.
#define PIN(x) asm volatile("" : "+v"(x))
// nothing moves across this point in instruction scheduling
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// LDS reads issued by hand (the compiler does not track them: the consumer waits with an explicit s_waitcnt that takes the
// destination registers as operands, so nothing that uses them can move above it)
__device__ __forceinline__ unsigned lds_addr(const void *ptr) { return (unsigned)(uintptr_t)ptr; }   // low half of a generic LDS pointer
#define lds_read4(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define lds_read_u16(dst, addr) asm volatile("ds_read_u16 %0, %1" : "=v"(dst) : "v"(addr))

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kT4 = 256, kW = 64, kTS = kW + 4, kSlice = 16, kChunk4 = 64, kRows = 32;

#ifdef NGPDE_STAMPS
// diagnostic build only (tools/stamps_edge64.py): phase timestamps of one steady-state tile per workgroup, [n_blocks][16] words
unsigned long long *g_edge64_stamps = nullptr;
#define E64_STAMP(k)                                                                                          \
  do {                                                                                                        \
    if (threadIdx.x == 0 && p.stamps && stamp_tile) {                                                         \
      p.stamps[(size_t)blockIdx.x * 16 + (k)] = clock64();                                                    \
      if ((k) == 0 || (k) == 13) p.stamps[(size_t)blockIdx.x * 16 + 14 + ((k) ? 1 : 0)] = wall_clock64();     \
    }                                                                                                         \
  } while (0)
#else
#define E64_STAMP(k)
#endif

struct EdgeMlp64K {
  const int4 *sched;
  const int2 *halo;
  const uint8_t *slots;
  int n_tiles, halo_rows, aggr;
  const float *P, *Q, *wt, *bias;
  float *out;
#ifdef NGPDE_STAMPS
  unsigned long long *stamps;
#endif
};

struct Meta64 {
  int4 sc0, sc1;       // schedule rows g16 and g16 + 16
  unsigned sw0, sw1;   // slot word (q & 7) of those rows
  int he[6];           // node ids of halo rows g16 + 16 k
};
struct Rows64 {
  float px[8];   // P rows g16 and g16 + 16
  float4 hv[6];
};

template <int ACT1, int ACT2>
__global__ __launch_bounds__(kT4, 2) void edge_mlp64_fwd_kernel(const EdgeMlp64K p) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  float *ldsQ = dyn;                                              // [halo_rows + 1][kTS]
  float *ldsP = ldsQ + (size_t)(p.halo_rows + 1) * kTS;           // [32][kTS]
  float *ldsWt = ldsP + kRows * kTS;                              // [64 out][kTS]   W2^T
  float *ldsMsg0 = ldsWt + kW * kTS;                              // [2][64 edges][kTS]: the messages of chunk c go to buffer c & 1, so
                                                                  // that one barrier per chunk orders them against the reduction
  __shared__ int ldsOff[kRows + 1];
  __shared__ __attribute__((aligned(16))) unsigned ldsSlots[kRows * 8];
  __shared__ uint16_t ldsEdge[kRows * kSlotWidth];   // tile edge k -> {row of the tile, halo slot << 8}
  __shared__ __attribute__((aligned(16))) float ldsBias[kW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g16 = tid >> 4, q = tid & 15;           // staging / reduction role: rows g16 and g16 + 16, features 4q .. 4q + 3
  const int ei = lane & 15, kq = lane >> 4;         // MFMA role
  const int zero_slot = p.halo_rows;

  const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
  const int range_len = p.n_tiles / 8 + (xcd < p.n_tiles % 8 ? 1 : 0);
  const int range_lo = xcd * (p.n_tiles / 8) + min(xcd, p.n_tiles % 8);

  auto fetch_meta = [&](int tile, Meta64 &m) {
    const size_t row = (size_t)tile * kTileRows + g16;
    m.sc0 = p.sched[row];
    m.sc1 = p.sched[row + 16];
    m.sw0 = reinterpret_cast<const unsigned *>(p.slots)[row * 8 + (q & 7)];
    m.sw1 = reinterpret_cast<const unsigned *>(p.slots)[(row + 16) * 8 + (q & 7)];
#pragma unroll
    for (int k = 0; k < 6; ++k) m.he[k] = p.halo[(size_t)tile * kHaloCap + min(g16 + 16 * k, kHaloCap - 1)].x;
  };
  auto fetch_rows = [&](const Meta64 &m, Rows64 &r) {
    const float4 p0 = *reinterpret_cast<const float4 *>(p.P + (size_t)max(m.sc0.x, 0) * kW + 4 * q);
    const float4 p1 = *reinterpret_cast<const float4 *>(p.P + (size_t)max(m.sc1.x, 0) * kW + 4 * q);
    r.px[0] = p0.x; r.px[1] = p0.y; r.px[2] = p0.z; r.px[3] = p0.w;
    r.px[4] = p1.x; r.px[5] = p1.y; r.px[6] = p1.z; r.px[7] = p1.w;
#pragma unroll
    for (int k = 0; k < 6; ++k)
      r.hv[k] = (g16 + 16 * k < p.halo_rows) ? *reinterpret_cast<const float4 *>(p.Q + (size_t)m.he[k] * kW + 4 * q) : f4_zero();
  };

  // ---- once per workgroup: W2^T rows (output j, contiguous inputs), bias, the all-zero halo row
  {
    const int j = tid & 63, kg0 = tid >> 6;   // output column j, input quads kg0 + 4 ps
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int k = 4 * (kg0 + 4 * ps);
      *reinterpret_cast<float4 *>(&ldsWt[j * kTS + k]) =
          make_float4(p.wt[(size_t)k * kW + j], p.wt[(size_t)(k + 1) * kW + j], p.wt[(size_t)(k + 2) * kW + j], p.wt[(size_t)(k + 3) * kW + j]);
    }
    if (tid < kW) ldsBias[tid] = p.bias ? p.bias[tid] : 0.f;
    if (g16 == 0) *reinterpret_cast<float4 *>(&ldsQ[zero_slot * kTS + 4 * q]) = f4_zero();
  }

  // ---- the activation in stages: y = fin(x, tr2(mid(tr1(pre(x))))) with the two quarter-rate transcendentals (tr1, tr2) as
  // separate steps, so that the pipelined block below can put exactly one of them behind every MFMA; pre / mid / fin on pairs of
  // values (packed instructions).  Same operations in the same order as act_apply (device_utils.h): bitwise the same values.
  auto tr1 = [](auto act, float t) -> float {
    constexpr int A = decltype(act)::value;
    return (A == NGPDE_ACT_SWISH || A == NGPDE_ACT_TANH) ? __builtin_amdgcn_exp2f(t) : t;
  };
  auto tr2 = [](auto act, float d) -> float {
    constexpr int A = decltype(act)::value;
    return (A == NGPDE_ACT_SWISH || A == NGPDE_ACT_TANH) ? __builtin_amdgcn_rcpf(d) : d;
  };
  // the packed (two-value) forms of the non-transcendental steps: the same operations on both halves
  auto pre2 = [](auto act, f32x2 x) -> f32x2 {
    constexpr int A = decltype(act)::value;
    if (A == NGPDE_ACT_SWISH) return (-x) * 1.4426950408889634f;
    if (A == NGPDE_ACT_TANH) return (2.0f * x) * 1.4426950408889634f;
    return x;
  };
  auto mid2 = [](auto act, f32x2 e) -> f32x2 {
    constexpr int A = decltype(act)::value;
    return (A == NGPDE_ACT_SWISH || A == NGPDE_ACT_TANH) ? 1.0f + e : e;
  };
  auto fin2 = [](auto act, f32x2 x, f32x2 r) -> f32x2 {
    constexpr int A = decltype(act)::value;
    if (A == NGPDE_ACT_SWISH) return x * r;
    if (A == NGPDE_ACT_TANH) return 1.0f - 2.0f * r;
    if (A == NGPDE_ACT_RELU) return (f32x2){fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)};
    return x;
  };
  const std::integral_constant<int, ACT1> act1_c{};
  const std::integral_constant<int, ACT2> act2_c{};

  // permanent registers: the bias of the lane's 16 output features, the W2^T fragments of the first 16 input features
  f32x2 bias2[8];   // bias2[2 mt + h] = bias of features 16 mt + 4 kq + 2 h, + 1
  f32x4 w0[4];
  {
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const float4 b4 = *reinterpret_cast<const float4 *>(&ldsBias[16 * mt + 4 * kq]);
      bias2[2 * mt] = (f32x2){b4.x, b4.y}; bias2[2 * mt + 1] = (f32x2){b4.z, b4.w};
      const float4 w4 = *reinterpret_cast<const float4 *>(&ldsWt[(mt * 16 + ei) * kTS + 4 * kq]);
      w0[mt] = (f32x4){w4.x, w4.y, w4.z, w4.w};
    }
  }
  auto comp = [](const f32x4 &v, int r) -> float { return v[r]; };
  const unsigned wt_addr = lds_addr(ldsWt + ei * kTS + 4 * kq);   // W2^T fragment (mt, ct) at + (mt * 16 * kTS + 16 * ct) floats

  // ---- one pipelined block = 64 slots.  Slot s: MFMA s of the current slice (ct = s / 16, r = (s / 4) % 4, mt = s % 4: four
  // independent accumulator chains) and one stage of the activation of a pair of values -- pairs 0..7 (slots 0..31) are the
  // messages of the PREVIOUS slice (bias + act2, stored to LDS as each float4 completes), pairs 8..15 (slots 32..63) the first
  // activation of the NEXT slice (P[row] + Q[slot]).  LDS reads are issued by hand in fixed slots and collected by three waits
  // that each sit several slots behind the last issue: slot 0 the lane's edge word, 1..4 the W2^T fragments of input block 1,
  // wait at 7; 17..20 fragments of block 2, 21..24 the P / Q rows of feature blocks 0 and 1, wait at 31; 33..36 the rows of
  // feature blocks 2 and 3, 37..40 the fragments of block 3, wait at 47.  A sched_barrier after every slot and the pins keep
  // this order through instruction selection and scheduling.
  //   MF: the products run; PUB: there is a previous slice to publish; ASM: assemble the slice `it_next`.
  auto block = [&](auto mf_c, auto pub_c, auto asm_c, int it_next, int total, float *ldsMsg, f32x4 (&a)[4], f32x4 (&accp)[4]) __attribute__((always_inline)) {
    constexpr bool MF = decltype(mf_c)::value, PUB = decltype(pub_c)::value, ASM = decltype(asm_c)::value;
    f32x4 acc[4];
    f32x4 wf[4][4];
    f32x2 xs2[16], ts2[16];
    f32x4 pr[4], qr[4];
    unsigned eword = 0, p_addr = 0, q_addr = 0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) wf[0][mt] = w0[mt];
    static_for<0, 64>([&](auto s_c) __attribute__((always_inline)) {
      constexpr int s = decltype(s_c)::value;
      constexpr int ct = s >> 4, rr = (s >> 2) & 3, mt = s & 3;
      if (MF) {
        const f32x4 c0 = (ct == 0 && rr == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mt];
        acc[mt] = mfma16(comp(wf[ct][mt], rr), comp(a[ct], rr), c0);
        PIN(acc[mt]);
        if (s >= 1 && s <= 4) lds_read4(wf[1][s - 1], wt_addr + ((s - 1) * 16 * kTS + 16) * 4);
        if (s >= 17 && s <= 20) lds_read4(wf[2][s - 17], wt_addr + ((s - 17) * 16 * kTS + 32) * 4);
        if (s >= 37 && s <= 40) lds_read4(wf[3][s - 37], wt_addr + ((s - 37) * 16 * kTS + 48) * 4);
      }
      if (ASM) {   // the next slice's edge word {row of the tile, halo slot}, then its P / Q rows
        if (s == 0) {
          const int k = min(it_next * kChunk4 + wave * kSlice + ei, total - 1);
          lds_read_u16(eword, lds_addr(ldsEdge) + 2 * k);
        }
        if (s == 8) {
          p_addr = lds_addr(ldsP) + ((eword & 0xff) * kTS + 4 * kq) * 4;
          q_addr = lds_addr(ldsQ) + ((eword >> 8) * kTS + 4 * kq) * 4;
          PIN(p_addr);
          PIN(q_addr);
        }
        if (s == 21) lds_read4(pr[0], p_addr);
        if (s == 22) lds_read4(qr[0], q_addr);
        if (s == 23) lds_read4(pr[1], p_addr + 64);
        if (s == 24) lds_read4(qr[1], q_addr + 64);
        if (s == 33) lds_read4(pr[2], p_addr + 128);
        if (s == 34) lds_read4(qr[2], q_addr + 128);
        if (s == 35) lds_read4(pr[3], p_addr + 192);
        if (s == 36) lds_read4(qr[3], q_addr + 192);
      }
      // the three collection points
      if (s == 7) {
        if (MF && ASM) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[1][2]), "+v"(wf[1][3]), "+v"(eword));
        else if (MF) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[1][2]), "+v"(wf[1][3]));
        else if (ASM) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(eword));
      }
      if (s == 31) {
        if (MF) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[2][0]), "+v"(wf[2][1]), "+v"(wf[2][2]), "+v"(wf[2][3]));
        if (ASM) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pr[0]), "+v"(qr[0]), "+v"(pr[1]), "+v"(qr[1]));
      }
      if (s == 47) {
        if (MF) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[3][0]), "+v"(wf[3][1]), "+v"(wf[3][2]), "+v"(wf[3][3]));
        if (ASM) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pr[2]), "+v"(qr[2]), "+v"(pr[3]), "+v"(qr[3]));
      }
      {
        // values travel in PAIRS (two consecutive features of the lane): the non-transcendental steps are one packed instruction
        // per pair (v_pk_add_f32 / v_pk_mul_f32) -- with the matrix instruction unable to run beside the VALU every such
        // instruction is a matrix-pipe bubble.  Pair pp, slots 4 pp .. 4 pp + 3: {source + pre-scale, tr1(e0)} {tr1(e1)}
        // {mid, tr2(e0)} {tr2(e1)}; its last step (fin) opens the next pair's first slot.
        constexpr int pp = s >> 2, ph = s & 3;
        constexpr bool pub_pair = pp < 8;
        if constexpr (ph == 0 && pp > 0) {
          constexpr bool prev_pub = (pp - 1) < 8;
          if (prev_pub ? PUB : ASM) {
            xs2[pp - 1] = prev_pub ? fin2(act2_c, xs2[pp - 1], ts2[pp - 1]) : fin2(act1_c, xs2[pp - 1], ts2[pp - 1]);
            PIN(xs2[pp - 1]);
          }
        }
        if constexpr (PUB && pub_pair && ph == 1 && (pp & 1) == 0 && pp > 0) {   // float4 mt = pp / 2 - 1 of the messages is complete
          constexpr int m = pp / 2 - 1;
          *reinterpret_cast<float4 *>(&ldsMsg[(wave * kSlice + ei) * kTS + 16 * m + 4 * kq]) =
              make_float4(xs2[2 * m][0], xs2[2 * m][1], xs2[2 * m + 1][0], xs2[2 * m + 1][1]);
        }
        if (PUB && s == 33)
          *reinterpret_cast<float4 *>(&ldsMsg[(wave * kSlice + ei) * kTS + 48 + 4 * kq]) = make_float4(xs2[6][0], xs2[6][1], xs2[7][0], xs2[7][1]);
        if constexpr (pub_pair ? PUB : ASM) {
          if constexpr (ph == 0) {
            if constexpr (pub_pair) {
              constexpr int mt_ = pp >> 1, h_ = 2 * (pp & 1);
              xs2[pp] = (f32x2){accp[mt_][h_], accp[mt_][h_ + 1]} + bias2[pp];
              ts2[pp] = pre2(act2_c, xs2[pp]);
              ts2[pp][0] = tr1(act2_c, ts2[pp][0]);
            } else {
              constexpr int c_ = (pp - 8) >> 1, h_ = 2 * (pp & 1);
              xs2[pp] = (f32x2){pr[c_][h_], pr[c_][h_ + 1]} + (f32x2){qr[c_][h_], qr[c_][h_ + 1]};
              ts2[pp] = pre2(act1_c, xs2[pp]);
              ts2[pp][0] = tr1(act1_c, ts2[pp][0]);
            }
            PIN(xs2[pp]);
            PIN(ts2[pp]);
          } else if constexpr (ph == 1) {
            ts2[pp][1] = pub_pair ? tr1(act2_c, ts2[pp][1]) : tr1(act1_c, ts2[pp][1]);
            PIN(ts2[pp]);
          } else if constexpr (ph == 2) {
            ts2[pp] = pub_pair ? mid2(act2_c, ts2[pp]) : mid2(act1_c, ts2[pp]);
            ts2[pp][0] = pub_pair ? tr2(act2_c, ts2[pp][0]) : tr2(act1_c, ts2[pp][0]);
            PIN(ts2[pp]);
          } else {
            ts2[pp][1] = pub_pair ? tr2(act2_c, ts2[pp][1]) : tr2(act1_c, ts2[pp][1]);
            PIN(ts2[pp]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (ASM) {
      xs2[15] = fin2(act1_c, xs2[15], ts2[15]);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) a[ct] = (f32x4){xs2[8 + 2 * ct][0], xs2[8 + 2 * ct][1], xs2[9 + 2 * ct][0], xs2[9 + 2 * ct][1]};
    }
    if (MF) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) accp[mt] = acc[mt];
    }
  };
  const std::true_type yes{};
  const std::false_type no{};

  Meta64 meta;
  Rows64 rows;
  int jt = wg_in_xcd;
  if (jt < range_len) {
    fetch_meta(range_lo + jt, meta);
    fetch_rows(meta, rows);
  }

  for (; jt < range_len; jt += wgs_per_xcd) {
#ifdef NGPDE_STAMPS
    const bool stamp_tile = (jt == wg_in_xcd + 4 * wgs_per_xcd);   // a tile in steady state (the fifth of the workgroup)
#endif
    E64_STAMP(0);
    // ---- stage this tile (fetched under the previous tile's arithmetic)
    const int4 sc0 = meta.sc0, sc1 = meta.sc1;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int hh = g16 + 16 * k;
      if (hh < p.halo_rows) *reinterpret_cast<float4 *>(&ldsQ[hh * kTS + 4 * q]) = rows.hv[k];
    }
    *reinterpret_cast<float4 *>(&ldsP[g16 * kTS + 4 * q]) = make_float4(rows.px[0], rows.px[1], rows.px[2], rows.px[3]);
    *reinterpret_cast<float4 *>(&ldsP[(g16 + 16) * kTS + 4 * q]) = make_float4(rows.px[4], rows.px[5], rows.px[6], rows.px[7]);
    if (q == 0) {
      ldsOff[g16 + 1] = sc0.x >= 0 ? sc0.z : 0;
      ldsOff[g16 + 17] = sc1.x >= 0 ? sc1.z : 0;
      if (g16 == 0) ldsOff[0] = 0;
    }
    if (q < 8) {
      ldsSlots[g16 * 8 + q] = meta.sw0;
      ldsSlots[(g16 + 16) * 8 + q] = meta.sw1;
    }
    const int jn = jt + wgs_per_xcd;
    const bool has_next = jn < range_len;       // workgroup-uniform
    if (has_next) fetch_meta(range_lo + jn, meta);
    __syncthreads();
    if (tid < kRows) {   // inclusive scan of the 32 degrees inside wave 0
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

    E64_STAMP(1);
    const int n_it = (total + kChunk4 - 1) / kChunk4;
    auto reduce = [&](int it) {   // lane group g16 sums the messages of rows g16 and g16 + 16 that lie in chunk `it`, edge order
      const int c0 = it * kChunk4;
      const float *base = ldsMsg0 + (it & 1) * (kChunk4 * kTS) + 4 * q - c0 * kTS;
      auto row_sum = [&](int lo, int hi, float4 &racc) {
        int kk = max(lo, c0);
        const int end = min(hi, c0 + kChunk4);
        for (; kk + 4 <= end; kk += 4) {   // four independent LDS reads, added in edge order
          const float4 m0 = *reinterpret_cast<const float4 *>(base + kk * kTS), m1 = *reinterpret_cast<const float4 *>(base + (kk + 1) * kTS);
          const float4 m2 = *reinterpret_cast<const float4 *>(base + (kk + 2) * kTS), m3 = *reinterpret_cast<const float4 *>(base + (kk + 3) * kTS);
          racc = f4_add(f4_add(f4_add(f4_add(racc, m0), m1), m2), m3);
        }
        for (; kk < end; ++kk) racc = f4_add(racc, *reinterpret_cast<const float4 *>(base + kk * kTS));
      };
      row_sum(lo0, hi0, racc0);
      row_sum(lo1, hi1, racc1);
    };
    if (n_it > 0) {
      f32x4 a[4], accp[4];
      block(no, no, yes, 0, total, ldsMsg0, a, accp);          // a1 of slice 0
      E64_STAMP(2);
      block(yes, no, yes, 1, total, ldsMsg0, a, accp);         // products of slice 0 | a1 of slice 1
      E64_STAMP(3);
      if (has_next) fetch_rows(meta, rows);           // the next tile's rows: in flight across this tile's remaining arithmetic
      E64_STAMP(4);
      for (int it = 1; it < n_it; ++it) {
        // products of slice it | messages of slice it - 1 (-> buffer (it - 1) & 1) | a1 of slice it + 1
        block(yes, yes, yes, it + 1, total, ldsMsg0 + ((it - 1) & 1) * (kChunk4 * kTS), a, accp);
        if (it == 1) E64_STAMP(5);
        __syncthreads();
        if (it == 1) E64_STAMP(6);
        reduce(it - 1);   // (no second barrier: the next block writes the other buffer)
        if (it == 1) E64_STAMP(7);
        if (it == 1) E64_STAMP(8);
        if (it == 2) E64_STAMP(9);
      }
      E64_STAMP(10);
      block(no, yes, no, 0, total, ldsMsg0 + ((n_it - 1) & 1) * (kChunk4 * kTS), a, accp);   // messages of the last slice
      E64_STAMP(11);
      __syncthreads();
      reduce(n_it - 1);
      __syncthreads();
      E64_STAMP(12);
    } else if (has_next) {
      fetch_rows(meta, rows);
    }
    if (sc0.x >= 0) {
      const int deg = hi0 - lo0;
      if (p.aggr == NGPDE_AGGR_MEAN) racc0 = deg > 0 ? f4_scale(1.0f / (float)deg, racc0) : f4_zero();
      *reinterpret_cast<float4 *>(p.out + (size_t)sc0.x * kW + 4 * q) = racc0;
    }
    if (sc1.x >= 0) {
      const int deg = hi1 - lo1;
      if (p.aggr == NGPDE_AGGR_MEAN) racc1 = deg > 0 ? f4_scale(1.0f / (float)deg, racc1) : f4_zero();
      *reinterpret_cast<float4 *>(p.out + (size_t)sc1.x * kW + 4 * q) = racc1;
    }
    E64_STAMP(13);
  }
}

// ---- pullback of the same message path (edge_mlp_fused_bwd_kernel's algorithm, specialised) -------------------------------------
// Per 16-edge wave slice, all in registers: z1 = P[t] + Q[s], s1 = sigma(z1) once for a1 = act1(z1) AND act1'(z1); z2^T = W2^T
// a1^T + b2 (MFMA); dz2 = g[t] . act2'(z2); dW2 += a1^T dz2 (MFMA over the slice's edges, operands transposed through the wave's
// 4 KB of LDS); da1^T = W2 dz2^T (MFMA); dz1 = da1 . act1'(z1) -> dE (memory, once) and through LDS into the per-target sum dP.
// What differs from the general kernel: 4 waves per workgroup and 64-edge chunks (a 192-edge tile is three full chunks; the
// general kernel's 128-edge chunks leave half of its waves idle in every second one), TWO workgroups per CU whose phases drift
// apart (one's tile loads and reductions under the other's products), the incoming gradient rows read where they are used
// instead of staged (that is what lets two workgroups fit the LDS), compile-time widths and activations, and one sigmoid per
// value where activation and derivative both need it.
struct EdgeMlp64BwdK {
  const int4 *sched;
  const int2 *halo;
  const uint8_t *slots;
  int n_tiles, halo_rows, aggr;
  const float *P, *Q, *wt, *bias, *dout;
  float *dP, *dE, *partial;   // partial: [n_workgroups][65][64]  (row 64 = bias gradient)
  float *dqpart;              // DQ: [n_tiles][kDqStride][64] per-tile sums of dz1 by FOREIGN halo slot (the by-source sum, first half)
  float *dQ;                  // DQ: the sums of a tile's OWN slots (= its own rows) go straight to the node's row
};
constexpr int kDqSlots = 3;                 // halo slots per 16-lane group whose sums stay in registers (DQ)
constexpr int kDqStride = 16 * kDqSlots;    // = the largest halo the in-launch by-source sum takes (48 rows)

// a = act(z), d = act'(z) with the shared transcendental evaluated once (same operations as act_apply / act_deriv)
template <int ACT>
__device__ __forceinline__ void act_both(float z, float &a, float &d) {
  if (ACT == NGPDE_ACT_SWISH) {
    const float s = sigmoidf_(z);
    a = z * s;
    d = s * (1.0f + z * (1.0f - s));
  } else if (ACT == NGPDE_ACT_TANH) {
    const float t = tanhf_(z);
    a = t;
    d = 1.0f - t * t;
  } else if (ACT == NGPDE_ACT_RELU) {
    a = fmaxf(z, 0.0f);
    d = z > 0.f ? 1.0f : 0.0f;
  } else {
    a = z;
    d = 1.0f;
  }
}

// DQ (no per-edge first-layer term, halos of at most kDqStride rows: meshes such as BASELINE config 4's): the [E][64] array dz1
// is NOT written.  Its only reader would be the by-source sum dQ; instead each 16-lane group keeps the sums of dz1 over the edges of
// "its" halo slots (slot = the edge's source row in the tile's halo list) in registers -- the edges of a chunk that carry a slot
// are found by ballots over the chunk's slot ids and added in edge order, so the result is reproducible.  The sums of the tile's OWN
// slots (slot k < 32 = its k-th row: most of a mesh's edges stay inside their tile) are the node's row of dQ up to what other tiles
// add: they are stored there; the sums of FOREIGN slots go to one row per (tile, slot), and edge64_dq_combine_kernel adds them to
// the few nodes that other tiles reference (graph handle: halo_inverse), ascending by tile.  (Round 4 wrote every slot's row and
// summed all of them per node: 452 MB of traffic for the 134 MB result; now 134 + 3 x 25 MB.)
template <int ACT1, int ACT2, bool DQ>
__global__ __launch_bounds__(kT4, 2) void edge_mlp64_bwd_kernel(const EdgeMlp64BwdK p) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  float *ldsQ = dyn;                                              // [halo_rows + 1][kTS]
  float *ldsP = ldsQ + (size_t)(p.halo_rows + 1) * kTS;           // [32][kTS]
  float *ldsS = ldsP + kRows * kTS;                               // [64][kTS]  wave-private transposes, then dz1 of the chunk
  float *ldsWf = ldsS + kChunk4 * kTS;                            // [64 out][kTS]  W2^T   (an unpadded XOR-swizzled image of both was measured in
  float *ldsWb = ldsWf + kW * kTS;                                // [64 in][kTS]   W2     round 5: conflict share 0.375 -> 0.231, launch 932 -> 958 us)
  __shared__ int ldsOff[kRows + 1], ldsRs[kRows], ldsNode[kRows];
  __shared__ float ldsInv[kRows];
  __shared__ __attribute__((aligned(16))) unsigned ldsSlots[kRows * 8];
  __shared__ uint16_t ldsEdge[kRows * kSlotWidth];
  __shared__ __attribute__((aligned(16))) float ldsBias[kW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g16 = tid >> 4, q = tid & 15;
  const int ei = lane & 15, kq = lane >> 4;
  const int zero_slot = p.halo_rows;

  const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
  const int range_len = p.n_tiles / 8 + (xcd < p.n_tiles % 8 ? 1 : 0);
  const int range_lo = xcd * (p.n_tiles / 8) + min(xcd, p.n_tiles % 8);

  auto fetch_meta = [&](int tile, Meta64 &m) {
    const size_t row = (size_t)tile * kTileRows + g16;
    m.sc0 = p.sched[row];
    m.sc1 = p.sched[row + 16];
    m.sw0 = reinterpret_cast<const unsigned *>(p.slots)[row * 8 + (q & 7)];
    m.sw1 = reinterpret_cast<const unsigned *>(p.slots)[(row + 16) * 8 + (q & 7)];
#pragma unroll
    for (int k = 0; k < 6; ++k) m.he[k] = p.halo[(size_t)tile * kHaloCap + min(g16 + 16 * k, kHaloCap - 1)].x;
  };

  {   // W2^T and W2 rows, bias, the all-zero halo row
    const int j = tid & 63, kg0 = tid >> 6;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int k = 4 * (kg0 + 4 * ps);
      *reinterpret_cast<float4 *>(&ldsWf[j * kTS + k]) =
          make_float4(p.wt[(size_t)k * kW + j], p.wt[(size_t)(k + 1) * kW + j], p.wt[(size_t)(k + 2) * kW + j], p.wt[(size_t)(k + 3) * kW + j]);
      *reinterpret_cast<float4 *>(&ldsWb[j * kTS + k]) = *reinterpret_cast<const float4 *>(p.wt + (size_t)j * kW + k);
    }
    if (tid < kW) ldsBias[tid] = p.bias ? p.bias[tid] : 0.f;
    if (g16 == 0) *reinterpret_cast<float4 *>(&ldsQ[zero_slot * kTS + 4 * q]) = f4_zero();
  }

  // dW2 accumulators of this wave: tile (ct, mt) <-> rows 16 ct .. + 15 (inputs) x columns 16 mt .. + 15 (outputs)
  f32x4 accW[4][4];
  float4 dbacc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    dbacc[a] = f4_zero();
#pragma unroll
    for (int b = 0; b < 4; ++b) accW[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  Meta64 meta;
  int jt = wg_in_xcd;
  if (jt < range_len) fetch_meta(range_lo + jt, meta);

  for (; jt < range_len; jt += wgs_per_xcd) {
    const int4 sc0 = meta.sc0, sc1 = meta.sc1;
    {   // stage the tile: P rows and the distinct Q rows (the other workgroup of the CU computes meanwhile)
      const float4 p0 = *reinterpret_cast<const float4 *>(p.P + (size_t)max(sc0.x, 0) * kW + 4 * q);
      const float4 p1 = *reinterpret_cast<const float4 *>(p.P + (size_t)max(sc1.x, 0) * kW + 4 * q);
      float4 hv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k)
        hv[k] = (g16 + 16 * k < p.halo_rows) ? *reinterpret_cast<const float4 *>(p.Q + (size_t)meta.he[k] * kW + 4 * q) : f4_zero();
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
    float4 qacc[kDqSlots];
#pragma unroll
    for (int j = 0; j < kDqSlots; ++j) qacc[j] = f4_zero();
    __syncthreads();

    for (int c0 = 0; c0 < total; c0 += kChunk4) {
      const bool wave_on = c0 + wave * kSlice < total;   // wave-uniform
      const int k = c0 + wave * kSlice + ei;
      const bool valid = k < total;
      float *mine = ldsS + (size_t)(wave * kSlice) * kTS;          // this wave's 16 rows of the staging tile
      float4 dz1[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
      if (wave_on) {
        const unsigned ew = ldsEdge[valid ? k : 0];
        const int r = ew & 0xff, slot = valid ? (int)(ew >> 8) : zero_slot;
        const size_t pe = (size_t)(ldsRs[r] + (k - ldsOff[r]));
        // incoming gradient rows of the edge's target (g = dout / deg for mean): issued now, used behind the first product
        // (kept raw until then: scaled at the load, the compiler loads them one after the other into one register
        // quad, each waited for before the next is issued -- three exposed round trips per slice)
        float4 gz[4];
        const float *grow = p.dout + (size_t)ldsNode[r] * kW + 4 * kq;
        const float inv = valid ? ldsInv[r] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) gz[mt] = *reinterpret_cast<const float4 *>(grow + 16 * mt);
        float4 a1[4], d1[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int f = 16 * ct + 4 * kq;
          const float4 z = f4_add(*reinterpret_cast<const float4 *>(&ldsP[r * kTS + f]), *reinterpret_cast<const float4 *>(&ldsQ[slot * kTS + f]));
          act_both<ACT1>(z.x, a1[ct].x, d1[ct].x);
          act_both<ACT1>(z.y, a1[ct].y, d1[ct].y);
          act_both<ACT1>(z.z, a1[ct].z, d1[ct].z);
          act_both<ACT1>(z.w, a1[ct].w, d1[ct].w);
          if (!valid) a1[ct] = f4_zero();
        }
        // ---- z2 (transposed product), dz2 = g . act2'(z2).  The W2^T fragment of group (mt, ct) + 1 is asked for in front of the
        // four products of group (mt, ct): left to itself the compiler puts every ds_read right before its use and waits for it
        // (16 exposed LDS latencies per product; `SCHED_FENCE` keeps the order through scheduling)
        {
          const float *wl0 = ldsWf + ei * kTS + 4 * kq;
          float4 wq[2];
          float4 bq[2];
          wq[0] = *reinterpret_cast<const float4 *>(wl0);
          bq[0] = *reinterpret_cast<const float4 *>(&ldsBias[4 * kq]);
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const int mt = g >> 2, ct = g & 3;
            if (g + 1 < 16) wq[(g + 1) & 1] = *reinterpret_cast<const float4 *>(wl0 + ((g + 1) >> 2) * 16 * kTS + 16 * ((g + 1) & 3));
            if (ct == 0 && mt + 1 < 4) bq[(mt + 1) & 1] = *reinterpret_cast<const float4 *>(&ldsBias[16 * (mt + 1) + 4 * kq]);
            SCHED_FENCE();
            const float4 w4 = wq[g & 1];
            acc = mfma16(w4.x, a1[ct].x, acc);
            acc = mfma16(w4.y, a1[ct].y, acc);
            acc = mfma16(w4.z, a1[ct].z, acc);
            acc = mfma16(w4.w, a1[ct].w, acc);
            SCHED_FENCE();
            if (ct == 3) {
              const float4 b4 = bq[mt & 1];
              const float4 z2 = make_float4(acc[0] + b4.x, acc[1] + b4.y, acc[2] + b4.z, acc[3] + b4.w);
              gz[mt] = f4_mul(f4_scale(inv, gz[mt]), make_float4(dact_c<ACT2>(z2.x), dact_c<ACT2>(z2.y), dact_c<ACT2>(z2.z), dact_c<ACT2>(z2.w)));
              dbacc[mt] = f4_add(dbacc[mt], gz[mt]);
              acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          }
        }
        // ---- dW2 += a1^T dz2 over this wave's 16 edges: both operands transposed through the wave's LDS rows
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<float4 *>(&mine[ei * kTS + 16 * ct + 4 * kq]) = a1[ct];
        float a1T[4][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int sI = 0; sI < 4; ++sI) a1T[ct][sI] = mine[(4 * sI + kq) * kTS + 16 * ct + ei];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<float4 *>(&mine[ei * kTS + 16 * mt + 4 * kq]) = gz[mt];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          float dzT[4];
#pragma unroll
          for (int sI = 0; sI < 4; ++sI) dzT[sI] = mine[(4 * sI + kq) * kTS + 16 * mt + ei];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) accW[ct][mt] = mfma16(a1T[ct][sI], dzT[sI], accW[ct][mt]);
        }
        // ---- da1 (transposed product with W2), dz1 = da1 . act1'(z1); fragments one group ahead as above
        {
          const float *wl0 = ldsWb + ei * kTS + 4 * kq;
          float4 wq[2];
          wq[0] = *reinterpret_cast<const float4 *>(wl0);
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const int ct = g >> 2, mt = g & 3;
            if (g + 1 < 16) wq[(g + 1) & 1] = *reinterpret_cast<const float4 *>(wl0 + ((g + 1) >> 2) * 16 * kTS + 16 * ((g + 1) & 3));
            SCHED_FENCE();
            const float4 w4 = wq[g & 1];
            acc = mfma16(w4.x, gz[mt].x, acc);
            acc = mfma16(w4.y, gz[mt].y, acc);
            acc = mfma16(w4.z, gz[mt].z, acc);
            acc = mfma16(w4.w, gz[mt].w, acc);
            SCHED_FENCE();
            if (mt == 3) {
              dz1[ct] = f4_mul(make_float4(acc[0], acc[1], acc[2], acc[3]), d1[ct]);
              if (!DQ && valid) *reinterpret_cast<float4 *>(p.dE + pe * kW + 16 * ct + 4 * kq) = dz1[ct];   // (a non-temporal store here: +3 %)
              acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          }
        }
      }
      // ---- dz1 of the chunk -> LDS, lane group g16 sums the rows of targets g16 and g16 + 16 in edge order (= dP)
      {
        float *mine2 = ldsS + (size_t)(wave * kSlice) * kTS;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<float4 *>(&mine2[ei * kTS + 16 * ct + 4 * kq]) = dz1[ct];
      }
      __syncthreads();
      {
        const float *base = ldsS + 4 * q - c0 * kTS;
        for (int kk = max(lo0, c0); kk < min(hi0, c0 + kChunk4); ++kk) racc0 = f4_add(racc0, *reinterpret_cast<const float4 *>(base + kk * kTS));
        for (int kk = max(lo1, c0); kk < min(hi1, c0 + kChunk4); ++kk) racc1 = f4_add(racc1, *reinterpret_cast<const float4 *>(base + kk * kTS));
      }
      if constexpr (DQ) {
        // lane q of a group looks at the slot ids of edges c0 + 16 i + q of the chunk; a ballot per i gives every group the 16 match
        // bits of its own slot; the set bits, lowest first, are the chunk's edges with that source in edge order
        unsigned sl4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kk = c0 + 16 * i + q;
          sl4[i] = kk < total ? (unsigned)(ldsEdge[kk] >> 8) : 0xffffu;
        }
        const int gsh = 16 * (g16 & 3);
#pragma unroll
        for (int j = 0; j < kDqSlots; ++j) {
          const unsigned s = (unsigned)(g16 + 16 * j);
          unsigned long long m = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i) m |= ((__ballot(sl4[i] == s) >> gsh) & 0xffffull) << (16 * i);
          float4 acc = qacc[j];
          while (m) {   // (a two-edges-per-slot, three-slots-at-once form of this loop was measured: 1 075 against 988 us for the launch)
            const int kk = __builtin_ctzll(m);
            m &= m - 1;
            acc = f4_add(acc, *reinterpret_cast<const float4 *>(ldsS + kk * kTS + 4 * q));
          }
          qacc[j] = acc;
        }
      }
      __syncthreads();
    }
    if constexpr (DQ) {
#pragma unroll
      for (int j = 0; j < kDqSlots; ++j) {
        const int sl = g16 + 16 * j;
        if (j < 2) {   // own rows g16 / g16 + 16 of the tile
          const int node = j == 0 ? sc0.x : sc1.x;
          if (node >= 0) *reinterpret_cast<float4 *>(p.dQ + (size_t)node * kW + 4 * q) = qacc[j];
        } else if (sl < p.halo_rows) {
          *reinterpret_cast<float4 *>(p.dqpart + ((size_t)(range_lo + jt) * kDqStride + sl) * kW + 4 * q) = qacc[j];
        }
      }
    }
    if (p.dP) {
      if (sc0.x >= 0) *reinterpret_cast<float4 *>(p.dP + (size_t)sc0.x * kW + 4 * q) = racc0;
      if (sc1.x >= 0) *reinterpret_cast<float4 *>(p.dP + (size_t)sc1.x * kW + 4 * q) = racc1;
    }
  }

  // ---- fold the waves' dW2 / db2 accumulators into this workgroup's slab, wave by wave (fixed order), then write it out
  {
    float *slab = ldsS;                                            // [65][64] <= [64][kTS]
    __syncthreads();
    for (int idx = tid; idx < (kW + 1) * kW; idx += kT4) slab[idx] = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {   // db: sum the 16 edge lanes of each k-quarter inside the wave first
      float v[4] = {dbacc[mt].x, dbacc[mt].y, dbacc[mt].z, dbacc[mt].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) v[c] += __shfl_xor(v[c], o);
      }
      dbacc[mt] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    for (int w = 0; w < kT4 / 64; ++w) {
      if (wave == w) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(16 * ct + 4 * kq + r) * kW + 16 * mt + ei] += accW[ct][mt][r];
        if (ei == 0) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int o = 16 * mt + 4 * kq;
            slab[kW * kW + o] += dbacc[mt].x; slab[kW * kW + o + 1] += dbacc[mt].y;
            slab[kW * kW + o + 2] += dbacc[mt].z; slab[kW * kW + o + 3] += dbacc[mt].w;
          }
        }
      }
      __syncthreads();
    }
    float *dst = p.partial + (size_t)blockIdx.x * (kW + 1) * kW;
    for (int idx = tid; idx < (kW + 1) * kW; idx += kT4) dst[idx] = slab[idx];
  }
}

bool env_off(const char *name) {
  const char *e = std::getenv(name);
  return e && e[0] == '1';
}

}  // namespace

// The specialised launch applies to: P and Q present, no per-edge term, h1 = 64, one further Dense 64 => 64, nothing saved per
// edge, + / mean, and an activation pair that is instantiated below.  NGPDE_NO_EDGE64=1 keeps the general kernel (read per call:
// the tests switch it at run time).
bool edge_mlp64_fwd_applicable(const ngpde_graph *g, const EdgeMlpArgs &a) {
  if (env_off("NGPDE_NO_EDGE64")) return false;
  if (!a.P || !a.Q || a.Eterm || a.h1 != kW || a.n_tail != 1 || a.din[0] != kW || a.dout[0] != kW) return false;
  for (int l = 0; l < 4; ++l)
    if (a.save_z[l]) return false;
  if (a.aggr != NGPDE_AGGR_SUM && a.aggr != NGPDE_AGGR_MEAN) return false;
  const bool a1 = a.act1 == NGPDE_ACT_SWISH || a.act1 == NGPDE_ACT_RELU || a.act1 == NGPDE_ACT_TANH;
  const bool a2 = a.act[0] == a.act1 || a.act[0] == NGPDE_ACT_IDENTITY;
  return a1 && a2;
}

int32_t launch_edge_mlp64_fwd(const ngpde_graph *g, const EdgeMlpArgs &a, hipStream_t stream) {
  EdgeMlp64K k;
  k.sched = g->by_t.sched; k.halo = g->by_t.halo; k.slots = g->by_t.slots;
  k.n_tiles = (int)(g->n_sched / kTileRows); k.aggr = a.aggr;
  k.halo_rows = std::max<int>(kTileRows, std::min<int>(kHaloCap, g->by_t.max_halo));
  k.P = a.P; k.Q = a.Q; k.wt = a.wt[0]; k.bias = a.bias[0]; k.out = a.out;
#ifdef NGPDE_STAMPS
  k.stamps = g_edge64_stamps;
#endif
  const size_t lds = ((size_t)(k.halo_rows + 1) * kTS + (size_t)kRows * kTS + (size_t)kW * kTS + 2 * (size_t)kChunk4 * kTS) * sizeof(float);
  int per_xcd = std::max(1, std::min(lds + 4096 <= 80 * 1024 ? 64 : 32, (k.n_tiles + 7) / 8));
  if (const char *e = std::getenv("NGPDE_EDGE64_WGS_PER_XCD")) per_xcd = std::max(1, std::min(per_xcd, atoi(e)));   // diagnostic: fewer resident workgroups   // two persistent workgroups per CU (one when the halo
                                                                                                   // region is large), a multiple of the 8 XCDs
  const dim3 grid(8 * per_xcd), block(kT4);
  auto launch = [&](auto kernel) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, k);
    return hipSuccess;
  };
  hipError_t le;
  const bool same = a.act[0] == a.act1;
  switch (a.act1) {
    case NGPDE_ACT_SWISH: le = same ? launch(edge_mlp64_fwd_kernel<NGPDE_ACT_SWISH, NGPDE_ACT_SWISH>) : launch(edge_mlp64_fwd_kernel<NGPDE_ACT_SWISH, NGPDE_ACT_IDENTITY>); break;
    case NGPDE_ACT_RELU: le = same ? launch(edge_mlp64_fwd_kernel<NGPDE_ACT_RELU, NGPDE_ACT_RELU>) : launch(edge_mlp64_fwd_kernel<NGPDE_ACT_RELU, NGPDE_ACT_IDENTITY>); break;
    default: le = same ? launch(edge_mlp64_fwd_kernel<NGPDE_ACT_TANH, NGPDE_ACT_TANH>) : launch(edge_mlp64_fwd_kernel<NGPDE_ACT_TANH, NGPDE_ACT_IDENTITY>); break;
  }
  if (le != hipSuccess) return fail(NGPDE_ERR_HIP, "edge_mlp64_fwd_kernel: LDS request of %zu bytes refused: %s", lds, hipGetErrorString(le));
  NGPDE_LAUNCH_CHECK("edge_mlp64_fwd_kernel");
  return NGPDE_OK;
}

// dQ[node] += the partial rows of the OTHER tiles whose halo list holds the node (entries ascending by tile: fixed order); the node's
// own tile has stored its share.  One 16-lane group per listed node (the nodes no other tile references are not touched).
__global__ __launch_bounds__(256) void edge64_dq_combine_kernel(int n_listed, const int *__restrict__ nodes, const int *__restrict__ ptr,
                                                                const int *__restrict__ ent, const float *__restrict__ part, float *__restrict__ dQ) {
  const int k = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4), q = threadIdx.x & 15;
  if (k >= n_listed) return;
  const int lo = ptr[k], hi = ptr[k + 1];
  float4 *dst = reinterpret_cast<float4 *>(dQ + (size_t)nodes[k] * kW + 4 * q);
  float4 a = *dst;
  for (int e = lo; e < hi; ++e) a = f4_add(a, *reinterpret_cast<const float4 *>(part + (size_t)ent[e] * kW + 4 * q));
  *dst = a;
}

// ---- pullback launch.  Same conditions as the forward specialisation (the activation pairs instantiated below); workspace =
// one [65][64] slab per workgroup (edge_mlp64_bwd_grid).
static int edge64_bwd_grid(const ngpde_graph *g) {
  const int n_tiles = (int)(g->n_sched / kTileRows);
  return 8 * std::max(1, std::min(64, (n_tiles + 7) / 8));   // two persistent workgroups per CU, a multiple of the 8 XCDs
}
bool edge_mlp64_bwd_applicable(const ngpde_graph *g, const EdgeMlpBwdArgs &a) {
  if (env_off("NGPDE_NO_EDGE64")) return false;
  if (!a.P || !a.Q || a.Eterm || a.h1 != kW || a.n_tail != 1 || a.dw != kW) return false;
  if (a.aggr != NGPDE_AGGR_SUM && a.aggr != NGPDE_AGGR_MEAN) return false;
  const bool a1 = a.act1 == NGPDE_ACT_SWISH || a.act1 == NGPDE_ACT_RELU || a.act1 == NGPDE_ACT_TANH;
  const bool a2 = a.act2 == a.act1 || a.act2 == NGPDE_ACT_IDENTITY;
  return a1 && a2;
}
static size_t edge64_slab_bytes(const ngpde_graph *g) { return ((size_t)edge64_bwd_grid(g) * (kW + 1) * kW * sizeof(float) + 255) / 256 * 256; }
// the by-source sum inside the launch (no [E][64] array): no per-edge term, every halo within kDqStride rows
bool edge_mlp64_bwd_dq_in_launch(const ngpde_graph *g, const EdgeMlpBwdArgs &a) {
  if (env_off("NGPDE_EDGE64_NO_DQ")) return false;
  return a.Eterm == nullptr && a.dQ != nullptr && g->by_t.halo_ok && g->by_t.max_halo <= kDqStride;
}
size_t edge_mlp64_bwd_workspace(const ngpde_graph *g) {
  const size_t part = (g->by_t.halo_ok && g->by_t.max_halo <= kDqStride) ? (size_t)(g->n_sched / kTileRows) * kDqStride * kW * sizeof(float) : 0;
  return edge64_slab_bytes(g) + part + 256;
}

int32_t launch_edge_mlp64_bwd(const ngpde_graph *g, const EdgeMlpBwdArgs &a, hipStream_t stream) {
  EdgeMlp64BwdK k;
  k.sched = g->by_t.sched; k.halo = g->by_t.halo; k.slots = g->by_t.slots;
  k.n_tiles = (int)(g->n_sched / kTileRows); k.aggr = a.aggr;
  k.halo_rows = std::max<int>(kTileRows, std::min<int>(kHaloCap, g->by_t.max_halo));
  k.P = a.P; k.Q = a.Q; k.wt = a.wt; k.bias = a.bias; k.dout = a.dout;
  k.dP = a.dP; k.dE = a.dE; k.partial = (float *)a.workspace;
  const bool dq = edge_mlp64_bwd_dq_in_launch(g, a);
  const HaloInverse *hinv = nullptr;
  if (dq) {
    int32_t sti = graph_halo_inverse(g, kDqStride, &hinv);
    if (sti) return sti;
  }
  NGPDE_REQUIRE(dq || a.dE != nullptr || g->n_edges == 0, NGPDE_ERR_INVALID_ARGUMENT, "fused edge-MLP pullback: the [E][h1] buffer dE is required");
  k.dqpart = dq ? reinterpret_cast<float *>(reinterpret_cast<char *>(a.workspace) + edge64_slab_bytes(g)) : nullptr;
  k.dQ = dq ? a.dQ : nullptr;
  const size_t lds = ((size_t)(k.halo_rows + 1) * kTS + (size_t)kRows * kTS + (size_t)kChunk4 * kTS + 2 * (size_t)kW * kTS) * sizeof(float);
  int grid = (lds + 4096 <= 80 * 1024) ? edge64_bwd_grid(g) : std::max(8, edge64_bwd_grid(g) / 2);
  if (const char *e = std::getenv("NGPDE_EDGE64_WGS_PER_XCD")) grid = std::max(8, std::min(grid, 8 * atoi(e)));   // diagnostic: fewer resident workgroups
  auto launch = [&](auto kernel) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kT4), lds, stream, k);
    return hipSuccess;
  };
  hipError_t le;
  const bool same = a.act2 == a.act1;
#define NGPDE_E64B(A1, A2) (dq ? launch(edge_mlp64_bwd_kernel<A1, A2, true>) : launch(edge_mlp64_bwd_kernel<A1, A2, false>))
  switch (a.act1) {
    case NGPDE_ACT_SWISH: le = same ? NGPDE_E64B(NGPDE_ACT_SWISH, NGPDE_ACT_SWISH) : NGPDE_E64B(NGPDE_ACT_SWISH, NGPDE_ACT_IDENTITY); break;
    case NGPDE_ACT_RELU: le = same ? NGPDE_E64B(NGPDE_ACT_RELU, NGPDE_ACT_RELU) : NGPDE_E64B(NGPDE_ACT_RELU, NGPDE_ACT_IDENTITY); break;
    default: le = same ? NGPDE_E64B(NGPDE_ACT_TANH, NGPDE_ACT_TANH) : NGPDE_E64B(NGPDE_ACT_TANH, NGPDE_ACT_IDENTITY); break;
  }
#undef NGPDE_E64B
  if (le != hipSuccess) return fail(NGPDE_ERR_HIP, "edge_mlp64_bwd_kernel: LDS request of %zu bytes refused: %s", lds, hipGetErrorString(le));
  NGPDE_LAUNCH_CHECK("edge_mlp64_bwd_kernel");
  int32_t st;
  if ((st = launch_dense_weight_reduce(grid, kW, kW, k.partial, a.dwt, a.dbias, stream))) return st;
  if (dq) {
    if (hinv->n_listed > 0) {
      const int64_t threads = (int64_t)hinv->n_listed * 16;
      hipLaunchKernelGGL(edge64_dq_combine_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, hinv->n_listed, hinv->node, hinv->ptr,
                         hinv->ent, k.dqpart, a.dQ);
      NGPDE_LAUNCH_CHECK("edge64_dq_combine_kernel");
    }
    return NGPDE_OK;
  }
  if (a.dQ && (st = launch_edge_sum_by_source(g, kW, a.dE, a.dQ, stream))) return st;
  return NGPDE_OK;
}

}  // namespace ngpde

#ifdef NGPDE_STAMPS
extern "C" int32_t ngpde_debug_set_edge64_stamps(unsigned long long *buf) {
  ngpde::g_edge64_stamps = buf;
  return 0;
}
#endif
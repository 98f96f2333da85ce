// device_utils.h -- gfx950 device helpers: activations, CSR row aggregation, fp32 MFMA tiles.
#pragma once

#include <hip/hip_runtime.h>

#include "common.h"

namespace ngpde {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- activations (codes: ngpde_act_t) -------------------------------------------------------------

// Transcendentals on the hardware units: v_exp_f32 / v_log_f32 / v_rcp_f32 are 1-ulp quarter-rate instructions; the libm
// forms (expf + IEEE division, tanhf) cost ~40 VALU instructions per element and made swish the bottleneck of the fused
// edge kernel (32 activations per edge at C4).  Absolute error of every form below <= ~2e-7 on the value range an
// activation sees, two orders below the parity tolerance (1e-4 relative to max |y|).
__device__ __forceinline__ float fast_exp(float z) { return __builtin_amdgcn_exp2f(z * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float z) { return __builtin_amdgcn_rcpf(z); }
__device__ __forceinline__ float fast_log(float z) { return __builtin_amdgcn_logf(z) * 0.6931471805599453f; }
__device__ __forceinline__ float sigmoidf_(float z) { return fast_rcp(1.0f + fast_exp(-z)); }
__device__ __forceinline__ float tanhf_(float z) { return 1.0f - 2.0f * fast_rcp(1.0f + fast_exp(2.0f * z)); }

__device__ __forceinline__ float act_apply(int act, float z) {
  switch (act) {
    case NGPDE_ACT_RELU: return fmaxf(z, 0.0f);
    case NGPDE_ACT_TANH: return tanhf_(z);
    case NGPDE_ACT_SIGMOID: return sigmoidf_(z);
    case NGPDE_ACT_SWISH: return z * sigmoidf_(z);
    case NGPDE_ACT_GELU: {
      float u = 0.7978845608028654f * (z + 0.044715f * z * z * z);
      return 0.5f * z * (1.0f + tanhf_(u));
    }
    case NGPDE_ACT_LEAKYRELU: return z > 0.f ? z : 0.01f * z;
    case NGPDE_ACT_ELU: return z > 0.f ? z : fast_exp(z) - 1.0f;
    case NGPDE_ACT_SOFTPLUS: return z > 20.f ? z : fast_log(1.0f + fast_exp(z));
    default: return z;
  }
}

// derivative w.r.t. the pre-activation z (for relu / identity / leakyrelu the output y may be passed)
__device__ __forceinline__ float act_deriv(int act, float z) {
  switch (act) {
    case NGPDE_ACT_RELU: return z > 0.f ? 1.0f : 0.0f;
    case NGPDE_ACT_TANH: { float t = tanhf_(z); return 1.0f - t * t; }
    case NGPDE_ACT_SIGMOID: { float s = sigmoidf_(z); return s * (1.0f - s); }
    case NGPDE_ACT_SWISH: { float s = sigmoidf_(z); return s * (1.0f + z * (1.0f - s)); }
    case NGPDE_ACT_GELU: {
      float u = 0.7978845608028654f * (z + 0.044715f * z * z * z);
      float t = tanhf_(u);
      return 0.5f * (1.0f + t) + 0.5f * z * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 0.134145f * z * z);
    }
    case NGPDE_ACT_LEAKYRELU: return z > 0.f ? 1.0f : 0.01f;
    case NGPDE_ACT_ELU: return z > 0.f ? 1.0f : fast_exp(z);
    case NGPDE_ACT_SOFTPLUS: return sigmoidf_(z);
    default: return 1.0f;
  }
}

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_fma(float a, float4 v, float4 c) {
  return make_float4(fmaf(a, v.x, c.x), fmaf(a, v.y, c.y), fmaf(a, v.z, c.z), fmaf(a, v.w, c.w));
}
__device__ __forceinline__ float4 f4_scale(float a, float4 v) { return make_float4(a * v.x, a * v.y, a * v.z, a * v.w); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// the same sum as two packed adds (v_pk_add_f32), for files built with -fno-slp-vectorize where a loop is all additions and no MFMA
// runs beside it (the hub geometry's row sums)
__device__ __forceinline__ float4 f4_add_pk(float4 a, float4 b) {
  typedef float v2f_ __attribute__((ext_vector_type(2)));
  const v2f_ lo = (v2f_){a.x, a.y} + (v2f_){b.x, b.y}, hi = (v2f_){a.z, a.w} + (v2f_){b.z, b.w};
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
// vector forms: ONE uniform switch per call (a switch per element costs a scalar branch chain per element, which -- not the
// transcendentals -- dominated the VALU-side time of the fused edge kernels)
template <int ACT>
__device__ __forceinline__ float act_c(float z) { return act_apply(ACT, z); }   // ACT constant: the switch folds away
template <int ACT>
__device__ __forceinline__ float dact_c(float z) { return act_deriv(ACT, z); }

#define NGPDE_ACT_DISPATCH(act, F, ...)                              \
  switch (act) {                                                     \
    case NGPDE_ACT_RELU: F<NGPDE_ACT_RELU>(__VA_ARGS__); break;       \
    case NGPDE_ACT_TANH: F<NGPDE_ACT_TANH>(__VA_ARGS__); break;       \
    case NGPDE_ACT_SIGMOID: F<NGPDE_ACT_SIGMOID>(__VA_ARGS__); break; \
    case NGPDE_ACT_SWISH: F<NGPDE_ACT_SWISH>(__VA_ARGS__); break;     \
    case NGPDE_ACT_GELU: F<NGPDE_ACT_GELU>(__VA_ARGS__); break;       \
    case NGPDE_ACT_LEAKYRELU: F<NGPDE_ACT_LEAKYRELU>(__VA_ARGS__); break; \
    case NGPDE_ACT_ELU: F<NGPDE_ACT_ELU>(__VA_ARGS__); break;         \
    case NGPDE_ACT_SOFTPLUS: F<NGPDE_ACT_SOFTPLUS>(__VA_ARGS__); break; \
    default: F<NGPDE_ACT_IDENTITY>(__VA_ARGS__); break;               \
  }

template <int ACT, int N>
__device__ __forceinline__ void act_n(float4 (&z)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) z[i] = make_float4(act_c<ACT>(z[i].x), act_c<ACT>(z[i].y), act_c<ACT>(z[i].z), act_c<ACT>(z[i].w));
}
template <int ACT, int N>
__device__ __forceinline__ void dact_n(float4 (&z)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) z[i] = make_float4(dact_c<ACT>(z[i].x), dact_c<ACT>(z[i].y), dact_c<ACT>(z[i].z), dact_c<ACT>(z[i].w));
}
// z[i] <- act(z[i]) / act'(z[i]) for N float4, one uniform switch
template <int N>
__device__ __forceinline__ void f4n_act(int act, float4 (&z)[N]) {
  NGPDE_ACT_DISPATCH(act, act_n, z)
}
template <int N>
__device__ __forceinline__ void f4n_dact(int act, float4 (&z)[N]) {
  NGPDE_ACT_DISPATCH(act, dact_n, z)
}

__device__ __forceinline__ float4 f4_act(int act, float4 z) {
  float4 v[1] = {z};
  f4n_act<1>(act, v);
  return v[0];
}
__device__ __forceinline__ float4 f4_dact(int act, float4 z) {
  float4 v[1] = {z};
  f4n_dact<1>(act, v);
  return v[0];
}

// ---- CSR segmented aggregation of whole feature rows --------------------------------------------
// A "group" of LPR = D/4 adjacent lanes owns R rows; lane q of the group holds features 4q..4q+3 of
// each row as one float4 (a D=64 row = 256 B = 16 lanes x dwordx4: fully coalesced row gathers).
// The group's lanes first load up to LPR {col, coef} entries of a row with ONE coalesced 8-byte load
// each, then broadcast them with in-register lane shuffles while issuing R*U independent 16-byte row
// loads, so a wave keeps 4*R*U neighbour rows in flight.  No atomics: each destination row is summed
// by one group in CSR (= COO) order, so results are bitwise reproducible run to run.
//   acc[r] = c[row] * ( sum_e coef_e * X[col_e] + (self ? c[row] * X[row] : 0) )
template <int LPR, int R, int U>
__device__ __forceinline__ void aggregate_rows(const float4 *__restrict__ X4, const int *__restrict__ rowptr,
                                               const int2 *__restrict__ ent, const float *__restrict__ cnorm,
                                               int self_loops, int n_nodes, const int (&rows)[R], int q,
                                               float4 (&acc)[R]) {
  static_assert(LPR % U == 0, "unroll must divide the lanes per row");
  int rs[R], deg[R];
  int maxdeg = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool ok = rows[r] < n_nodes;
    rs[r] = ok ? rowptr[rows[r]] : 0;
    deg[r] = ok ? rowptr[rows[r] + 1] - rs[r] : 0;
    maxdeg = max(maxdeg, deg[r]);
    acc[r] = f4_zero();
  }
  for (int base = 0; base < maxdeg; base += LPR) {
    int ecol[R], ecf[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool ok = base + q < deg[r];
      int2 v = make_int2(0, 0);
      if (ok) v = ent[rs[r] + base + q];
      ecol[r] = v.x;
      ecf[r] = v.y;
    }
    const int nin = min(LPR, maxdeg - base);
    for (int e = 0; e < nin; e += U) {
      float4 v[R][U];
      float cf[R][U];
      bool vld[R][U];
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int src = e + u;
          int col = __shfl(ecol[r], src, LPR);
          cf[r][u] = __int_as_float(__shfl(ecf[r], src, LPR));
          vld[r][u] = (base + src) < deg[r];
          col = vld[r][u] ? col : 0;
          v[r][u] = X4[(size_t)col * LPR + q];
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (vld[r][u]) acc[r] = f4_fma(cf[r][u], v[r][u], acc[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (rows[r] < n_nodes) {
      const float ci = cnorm[rows[r]];
      if (self_loops) acc[r] = f4_fma(ci, X4[(size_t)rows[r] * LPR + q], acc[r]);
      acc[r] = f4_scale(ci, acc[r]);
    }
  }
}

// streaming (non-temporal) 16-byte store for outputs far larger than the L2s that a later launch reads (pair / chain Dense
// forwards: 143 -> 132 us and 173 -> 168 us at config 4; the pullback kernels' gradient arrays, consumed by the next launch, were
// 0-3 % slower with it and keep plain stores)
__device__ __forceinline__ void nt_store4(float *ptr, float4 v) {
  typedef float v4f_ __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store((v4f_){v.x, v.y, v.z, v.w}, reinterpret_cast<v4f_ *>(ptr));
}

// s_waitcnt vmcnt(0) as an instruction the compiler's wait-count pass tracks.  An asm s_waitcnt is invisible to it: it keeps
// counting an LDS-DMA (global_load_lds) as pending and drains vmcnt again at the next barrier or LDS access -- with whatever
// loads and stores are in flight by then.  (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15.)
__device__ __forceinline__ void wait_vmcnt0() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" ::: "memory");
}

// row / d for a divisor that is fixed for the launch: m = div_magic(d) once, then a high multiply and one correction instead of the ~ 30
// VALU instructions of a 32-bit division.  With m = floor(2^32 / d) (2^32 - 1 for d = 1) the estimate mulhi(r, m) is the quotient or one
// below it for every r < 2^32.  (The narrow features of the streaming Dense launches -- coordinates per node, theta per trajectory --
// are fetched per row and tile: 16 divisions per thread and tile in dense_stream64_bwd_kernel.)
__device__ __forceinline__ uint32_t div_magic(uint32_t d) { return d <= 1 ? 0xFFFFFFFFu : (uint32_t)(0x100000000ull / d); }
__device__ __forceinline__ uint32_t fast_div(uint32_t r, uint32_t d, uint32_t m) {
  uint32_t q = __umulhi(r, m);
  if (r - q * d >= d) ++q;
  return q;
}

// The streaming Dense launches run two workgroups per CU whose tiles alternate a matrix-pipe phase and a memory phase.  Started
// together the two fall into step (both on the pipe at half rate, then both on the memory system) and a launch costs the SUM of its
// MFMA time and its memory time; the second workgroup of every CU therefore starts `cycles` late.
__device__ __forceinline__ void dephase_second_half(int cycles) {
  if (cycles > 0 && blockIdx.x >= (gridDim.x >> 1)) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(16);
  }
}

// ---- fp32 MFMA -------------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16]; lane l supplies A[l&15][l>>4], B[l>>4][l&15];
// result register r of lane l is D[4*(l>>4) + r][l&15].  Exact fp32 (bitwise an fmaf chain), 256 FLOP/clk/CU.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

}  // namespace ngpde

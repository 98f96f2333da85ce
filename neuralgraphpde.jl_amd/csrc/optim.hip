// optim.hip -- optimiser step on the FLAT parameter vector: the reference's tutorials keep
// the parameters as one ComponentArray and call Optimisers.update on it (/root/reference/docs/src/tutorials/
// graph_node.md:90,122-129 Adam; VMH.md:97 Rprop).  One launch right behind the gradient all-reduce on the same stream;
// the 1/world averaging of the reduced gradient is folded into the kernel (grad_scale).
#include <algorithm>

#include "common.h"
#include "device_utils.h"

namespace ngpde {
namespace {

// [UPSTREAM Optimisers.jl Adam]: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
//                               x -= eta * (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps)
__global__ void adam_kernel(int64_t n, float *__restrict__ x, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, float eta, float b1, float b2, float eps, float c1, float c2, float gs) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = gs * g[i];
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    x[i] -= mi / c1 / (sqrtf(vi / c2) + eps) * eta;
  }
}

// [UPSTREAM Optimisers.jl Rprop]: per-element step size grows by l2 while the gradient keeps its sign, shrinks by l1 when it
// flips (and that step is skipped: the remembered gradient becomes 0); x -= step * sign(remembered gradient)
__global__ void rprop_kernel(int64_t n, float *__restrict__ x, const float *__restrict__ g, float *__restrict__ gprev,
                             float *__restrict__ step, float l1, float l2, float smin, float smax, float gs) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = gs * g[i], p = gprev[i] * gi;
    float s = step[i];
    s = p > 0.f ? fminf(s * l2, smax) : (p < 0.f ? fmaxf(s * l1, smin) : s);
    const float keep = p < 0.f ? 0.f : gi;
    step[i] = s;
    gprev[i] = keep;
    x[i] -= s * (keep > 0.f ? 1.f : (keep < 0.f ? -1.f : 0.f));
  }
}

// out = c_self * base + sum_k coef[k] * term[k]: the Runge-Kutta stage input u + dt sum_j a_ij k_j, the step update, and the
// combinations of the discrete adjoint, for a right-hand side whose stages are evaluated by arbitrary layers.  float4 path when
// everything is 16-byte aligned.
struct CombK {
  const float *term[8];
  float coef[8];
};
// `out` may alias `base` or a term (in-place accumulation of parameter cotangents: element i is read, then written, by one thread):
// no __restrict__ on them
template <int N, class T>
__global__ void rk_combine_kernel(int64_t count, float c_self, const T *base, const CombK k, T *out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    T t[N > 0 ? N : 1];
#pragma unroll
    for (int j = 0; j < N; ++j) t[j] = reinterpret_cast<const T *>(k.term[j])[i];   // all loads first
    T v;
    if constexpr (sizeof(T) == 16) {
      v = base ? f4_scale(c_self, base[i]) : f4_zero();
#pragma unroll
      for (int j = 0; j < N; ++j) v = f4_fma(k.coef[j], t[j], v);
    } else {
      v = base ? c_self * base[i] : 0.f;
#pragma unroll
      for (int j = 0; j < N; ++j) v = fmaf(k.coef[j], t[j], v);
    }
    out[i] = v;
  }
}

// acc[k][i] += g[k][i] for up to kManyMax small arrays in one launch (blockIdx.y = array): the parameter cotangents of one
// right-hand-side pullback added to their accumulators (out aliases base: a thread reads and writes element i only)
constexpr int kManyMax = 24;
struct ManyK {
  float *acc[kManyMax];
  const float *g[kManyMax];
  int64_t count[kManyMax];
};
__global__ void accumulate_many_kernel(const ManyK k) {
  const int a = blockIdx.y;
  float *acc = k.acc[a];
  const float *g = k.g[a];
  const int64_t n = k.count[a];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc[i] = fmaf(1.0f, g[i], 1.0f * acc[i]);
}

template <class T>
void launch_rk_combine(int n_terms, int64_t count, float c_self, const float *base, const CombK &k, float *out, hipStream_t stream) {
  const dim3 grid((unsigned)std::min<int64_t>((count + 255) / 256, 4096)), block(256);
  const T *b = reinterpret_cast<const T *>(base);
  T *o = reinterpret_cast<T *>(out);
  switch (n_terms) {
    case 0: hipLaunchKernelGGL((rk_combine_kernel<0, T>), grid, block, 0, stream, count, c_self, b, k, o); break;
    case 1: hipLaunchKernelGGL((rk_combine_kernel<1, T>), grid, block, 0, stream, count, c_self, b, k, o); break;
    case 2: hipLaunchKernelGGL((rk_combine_kernel<2, T>), grid, block, 0, stream, count, c_self, b, k, o); break;
    case 3: hipLaunchKernelGGL((rk_combine_kernel<3, T>), grid, block, 0, stream, count, c_self, b, k, o); break;
    case 4: hipLaunchKernelGGL((rk_combine_kernel<4, T>), grid, block, 0, stream, count, c_self, b, k, o); break;
    case 5: hipLaunchKernelGGL((rk_combine_kernel<5, T>), grid, block, 0, stream, count, c_self, b, k, o); break;
    case 6: hipLaunchKernelGGL((rk_combine_kernel<6, T>), grid, block, 0, stream, count, c_self, b, k, o); break;
    case 7: hipLaunchKernelGGL((rk_combine_kernel<7, T>), grid, block, 0, stream, count, c_self, b, k, o); break;
    default: hipLaunchKernelGGL((rk_combine_kernel<8, T>), grid, block, 0, stream, count, c_self, b, k, o); break;
  }
}

}  // namespace
}  // namespace ngpde

using namespace ngpde;

extern "C" {

int32_t ngpde_rk_stage_combine(int64_t count, float c_self, const float *base, int32_t n_terms, const float *const *terms,
                               const float *coefs, float *out, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(count >= 0 && n_terms >= 0 && n_terms <= 8, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_rk_stage_combine: count >= 0 and 0 <= n_terms <= 8 required (got %lld, %d)", (long long)count, n_terms);
  if (count == 0) return NGPDE_OK;
  NGPDE_REQUIRE(out && (n_terms == 0 || (terms && coefs)), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_rk_stage_combine: NULL argument");
  CombK k;
  uintptr_t bits = reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(base);
  for (int j = 0; j < 8; ++j) {
    k.term[j] = j < n_terms ? terms[j] : nullptr;
    k.coef[j] = j < n_terms ? coefs[j] : 0.f;
    NGPDE_REQUIRE(j >= n_terms || terms[j], NGPDE_ERR_INVALID_ARGUMENT, "ngpde_rk_stage_combine: term %d is NULL", j);
    bits |= reinterpret_cast<uintptr_t>(k.term[j]);
  }
  if (count % 4 == 0 && (bits & 15) == 0) launch_rk_combine<float4>(n_terms, count / 4, c_self, base, k, out, (hipStream_t)stream);
  else launch_rk_combine<float>(n_terms, count, c_self, base, k, out, (hipStream_t)stream);
  NGPDE_LAUNCH_CHECK("rk_combine_kernel");
  return NGPDE_OK;
}

int32_t ngpde_accumulate_many(int32_t n_arrays, float *const *acc, const float *const *g, const int64_t *counts, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(n_arrays >= 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_accumulate_many: n_arrays < 0");
  if (n_arrays == 0) return NGPDE_OK;
  NGPDE_REQUIRE(acc && g && counts, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_accumulate_many: NULL argument");
  for (int a0 = 0; a0 < n_arrays; a0 += kManyMax) {
    ManyK k{};
    const int m = std::min(kManyMax, n_arrays - a0);
    int64_t longest = 0;
    for (int a = 0; a < m; ++a) {
      NGPDE_REQUIRE(counts[a0 + a] >= 0 && (counts[a0 + a] == 0 || (acc[a0 + a] && g[a0 + a])), NGPDE_ERR_INVALID_ARGUMENT,
                    "ngpde_accumulate_many: array %d is NULL or has a negative count", a0 + a);
      k.acc[a] = acc[a0 + a]; k.g[a] = g[a0 + a]; k.count[a] = counts[a0 + a];
      longest = std::max(longest, counts[a0 + a]);
    }
    if (longest == 0) continue;
    const dim3 grid((unsigned)std::min<int64_t>((longest + 255) / 256, 1024), (unsigned)m);
    hipLaunchKernelGGL(accumulate_many_kernel, grid, dim3(256), 0, (hipStream_t)stream, k);
    NGPDE_LAUNCH_CHECK("accumulate_many_kernel");
  }
  return NGPDE_OK;
}

int32_t ngpde_adam_step(int64_t n, float *x, const float *grad, float *m, float *v, float eta, float beta1, float beta2,
                        float eps, int64_t step, float grad_scale, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(n >= 0 && step >= 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_adam_step: n >= 0 and step >= 1 required");
  if (n == 0) return NGPDE_OK;
  NGPDE_REQUIRE(x && grad && m && v, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_adam_step: NULL argument");
  const float c1 = 1.0f - powf(beta1, (float)step), c2 = 1.0f - powf(beta2, (float)step);
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, x, grad, m, v, eta, beta1, beta2, eps, c1, c2,
                     grad_scale);
  NGPDE_LAUNCH_CHECK("adam_kernel");
  return NGPDE_OK;
}

int32_t ngpde_rprop_step(int64_t n, float *x, const float *grad, float *grad_prev, float *step_size, float shrink, float grow,
                         float step_min, float step_max, float grad_scale, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(n >= 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_rprop_step: n < 0");
  if (n == 0) return NGPDE_OK;
  NGPDE_REQUIRE(x && grad && grad_prev && step_size, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_rprop_step: NULL argument");
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(rprop_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, x, grad, grad_prev, step_size, shrink, grow,
                     step_min, step_max, grad_scale);
  NGPDE_LAUNCH_CHECK("rprop_kernel");
  return NGPDE_OK;
}

}  // extern "C"

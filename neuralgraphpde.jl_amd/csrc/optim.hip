// optim.hip -- optimiser step on the FLAT parameter vector: the reference's tutorials keep
// the parameters as one ComponentArray and call Optimisers.update on it (/root/reference/docs/src/tutorials/
// graph_node.md:90,122-129 Adam; VMH.md:97 Rprop).  One launch right behind the gradient all-reduce on the same stream;
// the 1/world averaging of the reduced gradient is folded into the kernel (grad_scale).
#include "common.h"

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

}  // namespace
}  // namespace ngpde

using namespace ngpde;

extern "C" {

int32_t ngpde_adam_step(int64_t n, float *x, const float *grad, float *m, float *v, float eta, float beta1, float beta2,
                        float eps, int64_t step, float grad_scale, ngpde_stream_t stream) {
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

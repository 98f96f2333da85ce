// Diagnostic: issue rate of v_mfma_f32_16x16x4_f32 for ONE wave per SIMD -- a dependent chain (one accumulator) and independent
// accumulators (4 / 10 in rotation) -- in clock64() ticks (s_memtime) and in wall_clock64() ticks (s_memrealtime, 100 MHz), with one
// workgroup on the device and with every CU busy (clock under load).  DESIGN 5.4: the pair pullback's products phase shows 47 - 84
// clock64 ticks per MFMA where the ISA says 32 cycles (8 passes).
// build + run:  hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rate_kernel(float *out, long long *ticks, int n) {
  f32x4 acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int u = 0; u < 40 / NACC; ++u)
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int k = 1; k < NACC; ++k) s += acc[k];
  const long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) {
    ticks[blockIdx.x * 2] = c1 - c0;
    ticks[blockIdx.x * 2 + 1] = w1 - w0;
  }
}

template <int NACC, int WAVES>
static int run(int grid, float *out, long long *ticks) {
  const int n = 2000;
  hipLaunchKernelGGL((rate_kernel<NACC, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, out, ticks, n);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((rate_kernel<NACC, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, out, ticks, n);
  CK(hipDeviceSynchronize());
  std::vector<long long> h(2 * grid);
  CK(hipMemcpy(h.data(), ticks, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost));
  const double mfmas = (double)n * (40 / NACC) * NACC;
  printf("{\"workgroups\": %d, \"waves_per_simd\": %.1f, \"accumulators\": %d, \"clock64_per_mfma\": %.1f, \"ns_per_mfma\": %.2f, \"implied_clock64_GHz\": %.2f}\n", grid, WAVES / 4.0, NACC,
         h[0] / mfmas, h[1] * 10.0 / mfmas, (double)h[0] / (h[1] * 10.0));
  return 0;
}

int main() {
  float *out;
  long long *ticks;
  CK(hipMalloc(&out, 4096 * 512 * 4));
  CK(hipMalloc(&ticks, 4096 * 16));
  int rc = 0;
  for (int grid : {1, 256, 512}) {
    rc |= run<1, 4>(grid, out, ticks);
    rc |= run<4, 4>(grid, out, ticks);
    rc |= run<10, 4>(grid, out, ticks);
    rc |= run<1, 8>(grid, out, ticks);
    rc |= run<4, 8>(grid, out, ticks);
  }
  return rc;
}

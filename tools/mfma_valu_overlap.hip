// tools/mfma_valu_overlap.hip -- does a SIMD of gfx950 run VALU / transcendental instructions beside its matrix instructions?
// One wave per SIMD slot (grid = 256 CUs x 4 SIMDs x W waves), each wave runs N iterations of
//   mode 0: 16 independent MFMAs                      mode 1: 16 x (v_exp_f32 + 2 v_fma_f32)        mode 2: both, interleaved 1 : 1
//   modes 3-6: four plain FMAs per slot / one transcendental per slot, alone and beside the fp32 matrix instruction
// for the fp32 matrix instruction (v_mfma_f32_16x16x4_f32) and the bf16 one (v_mfma_f32_16x16x32_bf16).  If mode 2 takes
// max(mode 0, mode 1) the pipes overlap; if it takes the sum they share issue / datapath.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// The 16 slots are one asm block (no compiler scheduling, register moves or wait states of its own in the loop): the accumulators
// rotate over four independent chains, a transcendental's consumer sits one s_nop behind it (the gfx94x/95x hazard).
#define M32(i) "v_mfma_f32_16x16x4_f32 %" #i ", %8, %9, %" #i "\n"
#define MBF(i) "v_mfma_f32_16x16x32_bf16 %" #i ", %8, %9, %" #i "\n"
#define VAL(i) "v_exp_f32 %" #i ", %" #i "\n s_nop 0\n v_fma_f32 %" #i ", %" #i ", %10, %11\n v_fma_f32 %" #i ", %" #i ", %12, %13\n"
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %10, %11\n v_fma_f32 %" #i ", %" #i ", %12, %13\n v_fma_f32 %" #i ", %" #i ", %10, %11\n v_fma_f32 %" #i ", %" #i ", %12, %13\n"
#define EXP(i) "v_exp_f32 %" #i ", %" #i "\n s_nop 0\n"
#define SLOTS4(M, V) M(0) V(4) M(1) V(5) M(2) V(6) M(3) V(7)
#define NONE(i) ""
template <int MODE, int KIND>
__global__ __launch_bounds__(256) void k(int n, float *out) {
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  float v0 = 0.1f * threadIdx.x, v1 = 0.2f, v2 = 0.3f, v3 = 0.4f;
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  const float c1 = 0.25f, c2 = 0.125f, c3 = 0.5f, c4 = -0.01f;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
#define BODY(STR, A, B)                                                                                                    \
  asm volatile(STR STR STR STR                                                                                             \
               : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)                     \
               : "v"(A), "v"(B), "v"(c1), "v"(c2), "v"(c3), "v"(c4))
  for (int it = 0; it < n; ++it) {
    if (MODE == 0 && KIND == 0) BODY(SLOTS4(M32, NONE), a, b);
    if (MODE == 1) BODY(SLOTS4(NONE, VAL), a, b);
    if (MODE == 2 && KIND == 0) BODY(SLOTS4(M32, VAL), a, b);
    if (MODE == 0 && KIND == 1) BODY(SLOTS4(MBF, NONE), ab, bb);
    if (MODE == 2 && KIND == 1) BODY(SLOTS4(MBF, VAL), ab, bb);
    if (MODE == 3) BODY(SLOTS4(NONE, FMA), a, b);          // four plain VALU instructions per slot, alone
    if (MODE == 4) BODY(SLOTS4(M32, FMA), a, b);           // ... beside the fp32 matrix instruction
    if (MODE == 5) BODY(SLOTS4(NONE, EXP), a, b);          // one transcendental per slot, alone
    if (MODE == 6) BODY(SLOTS4(M32, EXP), a, b);           // ... beside the fp32 matrix instruction
  }
  float r = v0 + v1 + v2 + v3;
  r += acc0[0] + acc1[1] + acc2[2] + acc3[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE, int KIND>
float run(int waves_per_simd, int n, float *out) {
  const dim3 grid(256 * waves_per_simd), block(256);   // 4 waves per workgroup = one per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, KIND>), grid, block, 0, 0, 16, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, KIND>), grid, block, 0, 0, n, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *out;
  hipMalloc(&out, sizeof(float) * 256 * 8 * 256);
  const int n = 20000;
  for (int w : {1, 2, 4}) {
    const float f0 = run<0, 0>(w, n, out), f1 = run<1, 0>(w, n, out), f2 = run<2, 0>(w, n, out);
    const float b0 = run<0, 1>(w, n, out), b2 = run<2, 1>(w, n, out);
    const double per = 1e6 / (double)n / 16.0;   // ns per slot
    const float v4 = run<3, 0>(w, n, out), m4 = run<4, 0>(w, n, out), t1 = run<5, 0>(w, n, out), mt = run<6, 0>(w, n, out);
    printf("{\"waves_per_simd\": %d, \"ns_per_slot\": {\"fp32_mfma\": %.2f, \"valu_exp_2fma\": %.2f, \"fp32_mfma+valu\": %.2f, "
           "\"bf16_mfma\": %.2f, \"bf16_mfma+valu\": %.2f, \"4fma\": %.2f, \"fp32_mfma+4fma\": %.2f, \"exp\": %.2f, "
           "\"fp32_mfma+exp\": %.2f}}\n", w, f0 * per, f1 * per, f2 * per, b0 * per, b2 * per, v4 * per, m4 * per, t1 * per, mt * per);
  }
  return 0;
}

// Diagnostic: what does THIS memory system give a streaming kernel whose read : write mix is that of the node-level Dense launches of
// the edge layers?  DESIGN §5.4 prices those launches against "a plain a + b reaches 6.3 TB/s", which is two reads per write; the
// launches themselves are write-heavy or five-stream:
//     dense_pair_fwd      1 read,  2 writes  (X -> P, Q)
//     dense_chain_fwd<2>  2 reads, 1 - 3 writes (X0, X1 -> y [, z1, a1 when training])
//     dense_stream64_bwd  2 - 3 reads, 1 - 2 writes
//     dense_pair64_bwd    3 reads, 2 writes  (dP, dQ, X -> dX, z)
// This program measures the arithmetic-free ceiling of each mix: NR input arrays and NW output arrays of `rows` x 64 fp32 (default: BASELINE
// config 4's 524 288 rows = 134 MB per array), out_j = sum_i in_i (+ j), one float4 per thread and step, persistent grid-stride workgroups
// of 512 threads, plain or non-temporal stores.  Bytes moved / median of 20 launches by HIP events.  At the default size an array fits the
// 256 MB MALL beside one or two others (the layer itself streams ~ 20 such arrays per pass); 4 194 304 rows (1 GB per array) is the MALL-free figure.
// build + run:  hipcc -O3 --offload-arch=gfx950 tools/stream_mix.hip -o /tmp/stream_mix && /tmp/stream_mix [rows] [workgroups per CU] [dynamic LDS bytes]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Ptrs {
  const float4 *in[4];
  float4 *out[4];
};

template <int NR, int NW, bool NT, int UNROLL>
__global__ __launch_bounds__(512) void mix_kernel(Ptrs p, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 512 * UNROLL;
  for (size_t base = (size_t)blockIdx.x * 512 * UNROLL + threadIdx.x; base < n4; base += stride) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const size_t k = base + (size_t)u * 512;
        if (k < n4) {
          const float4 a = p.in[i][k];
          v[u].x += a.x; v[u].y += a.y; v[u].z += a.z; v[u].w += a.w;
        }
      }
    if (NW == 0) {   // read only: keep the loads alive (the condition never holds on zero-filled inputs)
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (v[u].x == 12345.f) p.out[0][base] = v[u];
    }
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const size_t k = base + (size_t)u * 512;
        if (k < n4) {
          const float4 o = make_float4(v[u].x + j, v[u].y, v[u].z, v[u].w);
          if (NT) {
            float *q = reinterpret_cast<float *>(p.out[j] + k);
            __builtin_nontemporal_store(o.x, q);
            __builtin_nontemporal_store(o.y, q + 1);
            __builtin_nontemporal_store(o.z, q + 2);
            __builtin_nontemporal_store(o.w, q + 3);
          } else {
            p.out[j][k] = o;
          }
        }
      }
  }
}

static int g_lds = 0;   // dynamic LDS per workgroup: an occupancy limiter (80 KB: two workgroups per CU, as the Dense launches run)
template <int NR, int NW, bool NT, int UNROLL>
static int run(const char *what, Ptrs p, size_t n4, int grid) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int rep = 0; rep < 23; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((mix_kernel<NR, NW, NT, UNROLL>), dim3(grid), dim3(512), g_lds, 0, p, n4);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (rep >= 3) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double med = ms[ms.size() / 2], bytes = (double)(NR + NW) * n4 * 16;
  printf("{\"mix\": \"%s\", \"reads\": %d, \"writes\": %d, \"stores\": \"%s\", \"float4_per_thread\": %d, \"MB\": %.0f, \"us\": %.1f, \"TB_per_s\": %.2f}\n", what, NR, NW,
         NT ? "nontemporal" : "plain", UNROLL, bytes / 1e6, med * 1e3, bytes / (med * 1e-3) / 1e12);
  fflush(stdout);
  return 0;
}

int main(int argc, char **argv) {
  const size_t rows = argc > 1 ? (size_t)atoll(argv[1]) : 524288;
  const int per_cu = argc > 2 ? atoi(argv[2]) : 4;
  const size_t n4 = rows * 16;
  g_lds = argc > 3 ? atoi(argv[3]) : 0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount * per_cu;
  Ptrs p;
  for (int i = 0; i < 4; ++i) {
    void *a, *b;
    CK(hipMalloc(&a, n4 * 16));
    CK(hipMalloc(&b, n4 * 16));
    CK(hipMemset(a, 0, n4 * 16));
    CK(hipMemset(b, 0, n4 * 16));
    p.in[i] = (const float4 *)a;
    p.out[i] = (float4 *)b;
  }
  printf("{\"device\": \"%s\", \"CUs\": %d, \"workgroups\": %d, \"rows\": %zu, \"MB_per_array\": %.0f, \"lds_bytes\": %d}\n", prop.name, prop.multiProcessorCount, grid, rows, n4 * 16 / 1e6, g_lds);
  int rc = 0;
  rc |= run<2, 1, false, 2>("a + b (the comparison DESIGN 5.4 quotes)", p, n4, grid);
  rc |= run<1, 1, false, 2>("copy", p, n4, grid);
  rc |= run<1, 0, false, 2>("read only", p, n4, grid);
  rc |= run<0, 1, false, 2>("write only", p, n4, grid);
  rc |= run<0, 1, true, 2>("write only", p, n4, grid);
  rc |= run<1, 2, false, 2>("dense_pair_fwd: X -> P, Q", p, n4, grid);
  rc |= run<1, 2, true, 2>("dense_pair_fwd: X -> P, Q", p, n4, grid);
  rc |= run<1, 2, true, 4>("dense_pair_fwd: X -> P, Q", p, n4, grid);
  rc |= run<2, 1, true, 2>("dense_chain_fwd<2>, inference", p, n4, grid);
  rc |= run<2, 3, true, 2>("dense_chain_fwd<2>, training (y, z1, a1)", p, n4, grid);
  rc |= run<2, 3, false, 2>("dense_chain_fwd<2>, training (y, z1, a1)", p, n4, grid);
  rc |= run<3, 2, true, 2>("dense_pair64_bwd / dense_stream64_bwd<2>", p, n4, grid);
  rc |= run<3, 2, false, 2>("dense_pair64_bwd / dense_stream64_bwd<2>", p, n4, grid);
  rc |= run<3, 2, true, 4>("dense_pair64_bwd / dense_stream64_bwd<2>", p, n4, grid);
  rc |= run<3, 1, true, 2>("dense_stream64_bwd<1>", p, n4, grid);
  rc |= run<4, 4, true, 2>("eight streams", p, n4, grid);
  return rc;
}

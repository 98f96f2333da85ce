// Diagnostic: what does one node of a dependent kernel chain cost on this GPU when the kernel does (almost) nothing?
// Captures N launches of a kernel whose workgroups read one dword and write one dword into a HIP graph and replays it:
// the time per node is launch gap + dispatch ramp + drain, the floor under every launch of the solver plan.
// build + run on the GPU box:  hipcc -O2 --offload-arch=gfx950 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void touch(const int *__restrict__ in, int *__restrict__ out) {
  if (threadIdx.x == 0) out[blockIdx.x] = in[blockIdx.x] + 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const int N = 1200;
  int *a = nullptr, *b = nullptr;
  CK(hipMalloc(&a, 4096 * sizeof(int)));
  CK(hipMalloc(&b, 4096 * sizeof(int)));
  CK(hipMemset(a, 0, 4096 * sizeof(int)));
  CK(hipMemset(b, 0, 4096 * sizeof(int)));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int shapes[][2] = {{512, 512}, {256, 1024}, {512, 256}, {256, 256}, {1, 64}};
  for (auto &sh : shapes) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int k = 0; k < N; ++k) hipLaunchKernelGGL(touch, dim3(sh[0]), dim3(sh[1]), 0, s, (k & 1) ? b : a, (k & 1) ? a : b);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("{\"grid\": %d, \"block\": %d, \"nodes\": %d, \"us_per_node\": %.3f}\n", sh[0], sh[1], N, best * 1000.f / N);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}

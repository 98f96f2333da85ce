// Host-only stand-in for <hip/hip_runtime.h>, for ONE purpose: building the HOST side of this repo's own graph builder
// (neuralgraphpde.jl_amd/csrc/graph.hip) with -fsanitize=address,undefined on a machine without a GPU (GPU AddressSanitizer is
// not available on the test pool).  "Device" memory is host memory, copies are
// memcpy, there are no kernels: every device-side entry of the library is out of reach of this build (tests/asan/graph_host_asan.cpp
// stubs the two it links against).  Test infrastructure; never part of the product library.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct ihipStream_t *hipStream_t;
typedef struct ihipEvent_t *hipEvent_t;
typedef struct ihipGraph *hipGraph_t;
typedef struct hipGraphExec *hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct float4 { float x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __restrict__
static inline hipError_t hipMalloc(void **p, size_t bytes) { *p = std::malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "host shim"; }

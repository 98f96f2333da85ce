// persistent_mem.h -- memory-access helpers shared by the persistent solver kernels (node_persistent.hip, node_vmh.hip, gat_fused.hip):
// write-through row stores, scalar-base global / streaming accesses, float4 selects.  Internal (anonymous namespace).
#pragma once

#include "common.h"
#include "device_utils.h"

namespace ngpde {
namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

// write-through row store: visible to sc1 loads of every XCD once drained.  Scalar base + 32-bit byte offset (one VGPR of
// address instead of a 64-bit pair per array: the adjoint kernel sits at the 128-VGPR edge)
// The hazard recogniser does not look inside inline assembly.  Two of gfx9's software-managed hazards apply to this instruction:
//  * "VALU writes SGPR -> VMEM reads that SGPR: 5 wait states" -- the base may have been re-materialised just before the statement
//    by an SGPR-spill reload (v_readlane_b32 sN, vM, lane: a VALU write of an SGPR; these kernels spill 20-50 SGPRs).  Without
//    the s_nop 4 the store can go out with the register pair's PREVIOUS content as its base: rows land in another array.  (Seen
//    as run-to-run differences in one slot of the interleaved kernels, and very likely round 2's unexplained GPU memory fault
//    when a tape store was moved next to a flag store.)
//  * "VMEM store of more than 64 bits followed by a write of its data VGPRs: 1 wait state" -- the s_nop 1 behind it.
__device__ __forceinline__ void store_sc1(float *base, unsigned byte_off, float4 v) {
  f4v t = {v.x, v.y, v.z, v.w};
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(byte_off), "v"(t), "s"(base) : "memory");
}
// base + byte offset with the base kept scalar, as GLOBAL-address-space accesses: through a generic pointer rebuilt from an
// integer they compile to flat_load / flat_store, which count on lgkmcnt as well, may alias LDS as far as the compiler knows,
// and so get s_waitcnt vmcnt(0) lgkmcnt(0) in front of them while it counts an LDS-DMA as pending
#define NGPDE_GLOBAL_AS __attribute__((address_space(1)))
__device__ __forceinline__ float4 ld4_g(const float *base, unsigned byte_off) {
  const f4v t = *reinterpret_cast<NGPDE_GLOBAL_AS const f4v *>(reinterpret_cast<uintptr_t>(base) + byte_off);
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st4_g(float *base, unsigned byte_off, float4 v) {
  const f4v t = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<NGPDE_GLOBAL_AS f4v *>(reinterpret_cast<uintptr_t>(base) + byte_off) = t;
}
__device__ __forceinline__ float4 ld4_stream_g(const float *base, unsigned byte_off) {   // touch-once rows (the tape): non-temporal
  const f4v t = __builtin_nontemporal_load(reinterpret_cast<NGPDE_GLOBAL_AS const f4v *>(reinterpret_cast<uintptr_t>(base) + byte_off));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st4_stream_g(float *base, unsigned byte_off, float4 v) {
  const f4v t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<NGPDE_GLOBAL_AS f4v *>(reinterpret_cast<uintptr_t>(base) + byte_off));
}
__device__ __forceinline__ unsigned ldu8_g(const uint8_t *base, unsigned byte_off) {
  return (unsigned)*reinterpret_cast<NGPDE_GLOBAL_AS const uint8_t *>(reinterpret_cast<uintptr_t>(base) + byte_off);
}
__device__ __forceinline__ void stu8_g(uint8_t *base, unsigned byte_off, uint8_t v) {
  *reinterpret_cast<NGPDE_GLOBAL_AS uint8_t *>(reinterpret_cast<uintptr_t>(base) + byte_off) = v;
}

// component-wise select: `cond ? a : b` on two float4 LVALUES is an lvalue select (clang picks an ADDRESS and copies), which
// keeps both operands in scratch memory
__device__ __forceinline__ float4 f4_sel(bool cnd, float4 a, float4 b) {
  return make_float4(cnd ? a.x : b.x, cnd ? a.y : b.y, cnd ? a.z : b.z, cnd ? a.w : b.w);
}

__device__ __forceinline__ float4 f4_nan() {
  const float n = __int_as_float(0x7fc00000);
  return make_float4(n, n, n, n);
}

}  // namespace
}  // namespace ngpde

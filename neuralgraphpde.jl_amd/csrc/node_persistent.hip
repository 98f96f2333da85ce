// node_persistent.hip -- the fixed-step neural graph ODE over Chain(GCNConv(64 => 64, relu), GCNConv(64 => 64, relu)) as TWO
// persistent launches (forward solve, discrete adjoint) instead of 1205 kernel launches replayed from HIP graphs.
// [caller of the hot path in the reference: docs/src/tutorials/graph_node.md:44-66, :78; layer: src/layers.jl:200-239]
//
// Why: at the BASELINE size (16 384 nodes = 512 tiles of 32 rows) every launch of the replayed plan is exactly one wave of
// co-resident workgroups, the tile -> workgroup map never changes, and each launch pays the platform's dependent-launch floor
// (1.87 us, tools/launch_floor.hip) plus a cold start in which halo lists, slot bytes, schedule entries and W are fetched again
// -- 1205 times.  Here a workgroup keeps its tile for the whole solve:
//   * in registers: the tile's own rows of u and of the six stage derivatives k_j (forward) / of lambda and the stage adjoints
//     U-bar_j (adjoint), the dW / db accumulators of both layers for the whole adjoint (no slab read-modify-write), slot bytes,
//     halo node ids, c;
//   * in LDS: both W^T, the biases, the tile's own rows of the array being exchanged (halo slots 0..31);
//   * exchanged through memory: only the 32 rows a tile produces per phase (sc1 = write-through stores) and the <= 64 rows of
//     OTHER tiles its halo references (sc1 loads straight into LDS by LDS-DMA).
// A grid-wide barrier costs 9.2 us per phase on this chip (tools/persistent_floor.hip, mode 3); the dependency is local, so a
// tile waits only for the tiles its halo references: one flag word per tile holding the last finished phase, written by one
// lane after every wave has drained its stores (s_waitcnt vmcnt(0)) and the workgroup has met at a barrier, polled relaxed by
// one wave of the consumer.  Measured floor of that hand-off with this geometry: 2.3-2.5 us per phase, every payload word
// checked (mode 1 of the same tool; the launch-boundary form is 1.87 us + the cold start).  The wait list is the symmetric
// closure of "my halo references a row of yours" over BOTH directions of the graph: a tile that waits for its readers of the
// previous phase also may overwrite the array they read (the two exchanged arrays ping-pong).
// Every spin is bounded (abort word + ~2 s timeout); an aborted solve poisons its outputs with NaN and raises the plan's
// fault flag (ngpde_node_fault).  All workgroups must be co-resident: the host checks the grid against the occupancy of the
// kernels and otherwise keeps the replayed plan (node.hip).  Two persistent solves in flight on one device (two processes or
// two streams) can starve each other of residency: one solve at a time per device, NGPDE_NO_PERSISTENT=1 for anything else.
//
// Arithmetic is that of the pre-scaled replayed plan (gcn_fused.hip, PRE = true) operation for operation: same aggregation
// order (slot bytes, then the own row), same fp32 MFMA products, same stage combinations in the same order.
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "device_utils.h"
#include "gcn_tile.h"
#include "persistent_mem.h"

namespace ngpde {

namespace {

constexpr int PD = 64;
using PG = Geo<PD>;
static_assert(PG::R == 1 && PG::GROUPS == kTM && PG::LPR == 16, "one 16-lane group per tile row");
constexpr int kXhF = (kHaloCap + 1) * PD;   // halo region (floats), +1: the all-zero row
constexpr int kTileF = kTM * PG::TS;        // one 32-row MFMA operand / result tile
constexpr int kWF = PD * PG::TS;            // one transposed weight matrix
constexpr int kMaxTileRounds = 8;            // tile rounds: at most this many tiles per workgroup
constexpr int kNbrStride = 64;              // wait-list stride per tile; at most 63 entries (one lane of the polling wave each,
                                            // lane 63 watches the abort word)

#ifdef NGPDE_STAMPS
// diagnostic build only (tools/stamps_persistent.py): shader-clock stamps of thread 0 at 8 points of the first g_pst_max phases
unsigned long long *g_pst_base = nullptr;
int g_pst_max = 0;
#define NGPDE_PST_FIELD unsigned long long *stamps; int stamps_max;
#define NGPDE_PST(m, ph, k)                                                                                   \
  do {                                                                                                        \
    if (threadIdx.x == 0 && (m).stamps && (ph) <= (m).stamps_max)                                             \
      (m).stamps[((size_t)blockIdx.x * (m).stamps_max + ((ph) - 1)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define NGPDE_PST_FIELD
#define NGPDE_PST(m, ph, k)
#endif

struct TileCtx {
  int tid, lane, wave_u, grp, q, tile, node, hcount, wmax, my_nbr;
  bool valid;
  float ci;
  // kept in LDS, not in registers (the 16 lanes of a group would each hold the same 8 words): the 32 slot bytes of every row
  // [32][8] and the node ids of the 64 foreign halo slots
  const unsigned *lds_slots;
  const int *lds_hnode;
  const float *lds_w;   // weighted graphs (tile rounds only): the tile's slot weights [32][32], else unused
};
constexpr int kMetaF = kTM * 8 + 2 * kTM;   // floats of LDS the two tables take

struct TileMeta {
  const int2 *halo;
  const uint8_t *slots;
  const int4 *sched;
  const int2 *tile_info;
  const int *nbr;
  const float *slot_w;   // [n_sched][kSlotWidth] edge weights in slot order, or NULL (unweighted)
  unsigned *flags, *abort_word;
  int n_tiles;
  const uint8_t *of_pre;   // [n_tiles][8] own rounds per wave when slots / sched are a plan's own-first tables, else NULL
  int *stats;   // [n_tiles][2] (forward, adjoint): slot-phases of the last launch whose halo rows were gathered ahead of time
  // hub geometry (HUB kernels only; then `nbr` is [n_tiles][kHubNbr])
  const int *hub_halo;        // [n_tiles][kHubHalo] node ids, own rows first
  const uint8_t *hub_slots;   // [n_tiles][kHubList]
  const int2 *hub_rows;       // [n_sched] {start, length} of the row's list inside its tile's slot bytes
  const int4 *hub_info;       // [n_tiles] {halo count, list bytes, long rows, 0}
  const int4 *hub_sched;      // [n_sched] {node or -1, 0, 0, bits of c[node]}: the hub geometry's OWN tile partition (see hub_partition)
  const uint8_t *hub_long;    // [n_tiles][kTileRows] rows (0 .. 31) with more than kSlotWidth entries
  const float *hub_w;         // [n_tiles][kHubList] edge weights of the entries, or NULL (unweighted)
  NGPDE_PST_FIELD
};

__device__ __forceinline__ void tile_ctx_init(const TileMeta &m, TileCtx &c, float *lds_meta, int tile = -1) {
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.wave_u = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  c.grp = c.tid >> 4;
  c.q = c.tid & 15;
  c.tile = tile >= 0 ? tile : xcd_tile(blockIdx.x, m.n_tiles);
  const size_t pos = (size_t)c.tile * kTM + c.grp;
  const int4 sc = m.sched[pos];
  c.valid = sc.x >= 0;
  c.node = max(sc.x, 0);
  c.ci = c.valid ? __int_as_float(sc.w) : 0.f;
  unsigned *ls = reinterpret_cast<unsigned *>(lds_meta);
  int *lh = reinterpret_cast<int *>(lds_meta + kTM * 8);
  if (c.q < 8) ls[c.grp * 8 + c.q] = reinterpret_cast<const unsigned *>(m.slots)[pos * 8 + c.q];
  if (c.q >= 8 && c.q < 10) lh[c.grp + 32 * (c.q - 8)] = m.halo[(size_t)c.tile * kHaloCap + c.grp + 32 * (c.q - 7)].x;
  c.lds_slots = ls;
  c.lds_hnode = lh;
  c.hcount = __builtin_amdgcn_readfirstlane(m.tile_info[c.tile].x);
  int wm = c.valid ? sc.z : 0;
  wm = max(wm, __shfl_xor(wm, 16));
  wm = max(wm, __shfl_xor(wm, 32));
  c.wmax = __builtin_amdgcn_readfirstlane(wm);
  c.my_nbr = m.nbr[(size_t)c.tile * kNbrStride + c.lane];
}

// Tile rounds (node_*_persistentK_kernel): the tables of ALL the workgroup's tiles stay in LDS for the whole launch -- slot words,
// halo node ids, schedule entries, wait list, halo count per tile -- so that a turn sets its context up from LDS instead of
// re-reading five arrays from memory (one round trip and a barrier per turn).
constexpr int kMetaKF = kMetaF + kTM * 4 + kNbrStride + 4;   // floats per tile
// Weighted graphs (GCNConv's edge_weight, src/layers.jl:206-231) run on the tile-round kernels only: those keep ONE layer's W in LDS,
// which leaves room for the 4 KB of slot weights per tile behind the tile's tables -- for at most kMaxTileRoundsW tiles per workgroup.
constexpr int kMaxTileRoundsW = 3;
constexpr int kSlotWF = kTM * kSlotWidth;
template <bool WGT> constexpr int meta_stride() { return kMetaKF + (WGT ? kSlotWF : 0); }
template <bool WGT> constexpr int meta_tiles() { return WGT ? kMaxTileRoundsW : kMaxTileRounds; }
template <bool WGT = false>
__device__ __forceinline__ void tile_tables_to_lds(const TileMeta &m, int tile, float *base) {
  const int tid = threadIdx.x, grp = tid >> 4, q = tid & 15;
  const size_t pos = (size_t)tile * kTM + grp;
  unsigned *ls = reinterpret_cast<unsigned *>(base);
  int *lh = reinterpret_cast<int *>(base + kTM * 8);
  int4 *lsc = reinterpret_cast<int4 *>(base + kMetaF);
  int *ln = reinterpret_cast<int *>(base + kMetaF + kTM * 4);
  if (q < 8) ls[grp * 8 + q] = reinterpret_cast<const unsigned *>(m.slots)[pos * 8 + q];
  if (q >= 8 && q < 10) lh[grp + 32 * (q - 8)] = m.halo[(size_t)tile * kHaloCap + grp + 32 * (q - 7)].x;
  if (q == 10) lsc[grp] = m.sched[pos];
  if (tid < kNbrStride) ln[tid] = m.nbr[(size_t)tile * kNbrStride + tid];
  if (tid == kNbrStride) ln[kNbrStride] = m.tile_info[tile].x;
  if constexpr (WGT) {
    float2 *lw = reinterpret_cast<float2 *>(base + kMetaKF);
    lw[tid] = reinterpret_cast<const float2 *>(m.slot_w + (size_t)tile * kSlotWF)[tid];   // 512 threads x 2 floats = [32][32]
  }
}
template <bool WGT = false>
__device__ __forceinline__ void tile_ctx_from_lds(TileCtx &c, int tile, const float *base) {
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.wave_u = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  c.grp = c.tid >> 4;
  c.q = c.tid & 15;
  c.tile = tile;
  const int4 sc = reinterpret_cast<const int4 *>(base + kMetaF)[c.grp];
  const int *ln = reinterpret_cast<const int *>(base + kMetaF + kTM * 4);
  c.valid = sc.x >= 0;
  c.node = max(sc.x, 0);
  c.ci = c.valid ? __int_as_float(sc.w) : 0.f;
  c.lds_slots = reinterpret_cast<const unsigned *>(base);
  c.lds_hnode = reinterpret_cast<const int *>(base + kTM * 8);
  c.hcount = __builtin_amdgcn_readfirstlane(ln[kNbrStride]);
  int wm = c.valid ? sc.z : 0;
  wm = max(wm, __shfl_xor(wm, 16));
  wm = max(wm, __shfl_xor(wm, 32));
  c.wmax = __builtin_amdgcn_readfirstlane(wm);
  c.my_nbr = ln[c.lane];
  c.lds_w = WGT ? base + kMetaKF : nullptr;
}

// Wait until every tile of the wait list has finished phase ph - 1: wave 0 polls, one flag per lane (lanes 0..62) and the abort
// word on lane 63, ONE load per lane and round (a second dependent load per round would double the polling period, which is
// the granularity a published flag is seen with); everybody meets at the barrier.  Returns false when the solve was aborted.
// Bounded: after ~2 s of the 100 MHz counter (the whole solve takes ~6 ms) the wave raises the abort word itself.
__device__ __forceinline__ bool tile_wait(const TileMeta &m, const TileCtx &c, int ph, int *s_ok, const unsigned *flags) {
  if (ph <= 1) return true;
  if (c.wave_u == 0) {
    const unsigned need = (unsigned)(ph - 1);
    const unsigned *addr = (c.lane == 63) ? m.abort_word : (c.my_nbr >= 0 ? flags + 32 * c.my_nbr : nullptr);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    for (unsigned it = 1;; ++it) {
      unsigned f = need;
      if (addr) f = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__any((int)(c.lane == 63 && f != 0))) { ok = false; break; }              // somebody gave up
      if (__all((int)(c.lane == 63 || f >= need))) break;
      if ((it & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
        if (c.lane == 0) __hip_atomic_store(m.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);   // (2 / 4 / 8 measured: no difference beyond run-to-run noise; the polling period is not what a phase waits for)
    }
    if (c.lane == 0) *s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return *s_ok != 0;
}
__device__ __forceinline__ bool tile_wait(const TileMeta &m, const TileCtx &c, int ph, int *s_ok) { return tile_wait(m, c, ph, s_ok, m.flags); }
// The same wait with its first round of flag loads issued earlier by the caller (poll_issue: `f` holds wave 0's samples, in flight
// under whatever the workgroup did in between).  A workgroup that is level with its neighbours finds the flags in that sample
// and only meets at the barrier; one that runs ahead spins here exactly as long as it leads.  (Used by the interleaved adjoint,
// whose slot-phase is long enough for the flags to be there.  What the stamps of tools/stamps_interleaved.py say about the hand-off:
// a flag store is seen by a poll from another XCD ~3-4 k cycles (1.2-1.5 us) after it was issued, a poll or a gather is a
// ~1.5 k-cycle round trip, the drain in front of the flag ~1 k: with only two slots the ~2.4 us of hand-off exceed the ~1.7 us of
// work the other slot offers in the forward kernel -- wherever the look is put, the difference is waited for.)
__device__ __forceinline__ bool tile_wait_primed(const TileMeta &m, const TileCtx &c, int ph, int *s_ok, const unsigned *flags, unsigned f) {
  if (c.wave_u == 0) {
    const unsigned need = (unsigned)(ph - 1);
    const unsigned *addr = (c.lane == 63) ? m.abort_word : (c.my_nbr >= 0 ? flags + 32 * c.my_nbr : nullptr);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    for (unsigned it = 1;; ++it) {
      if (__any((int)(c.lane == 63 && f != 0))) { ok = false; break; }
      if (__all((int)(c.lane == 63 || f >= need))) break;
      if ((it & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
        if (c.lane == 0) __hip_atomic_store(m.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      f = (c.lane == 63) ? 0u : need;
      if (addr) f = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (c.lane == 0) *s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return *s_ok != 0;
}

// every storing wave drains, the workgroup meets, ONE lane publishes (Guideline 16, R1)
__device__ __forceinline__ void tile_publish(const TileCtx &c, int ph, unsigned *flags) {
  wait_vmcnt0();
  __syncthreads();
  if (c.tid == 0) __hip_atomic_store(flags + 32 * c.tile, (unsigned)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tile_publish(const TileMeta &m, const TileCtx &c, int ph) { tile_publish(c, ph, m.flags); }

// the rows of OTHER tiles this tile's halo references: memory -> LDS slots 32.., sc1 (the producers stored them write-through
// in the previous phase; sc1 loads bypass this CU's L1, which may hold the same addresses from two phases ago)
__device__ __forceinline__ void tile_gather_foreign(const TileCtx &c, const float *X, float *ldsXh) {
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (4 * c.wave_u + 32 * (k + 1) < c.hcount) {   // wave-uniform: a wave's four groups stage four consecutive slots
      const unsigned off = (unsigned)c.lds_hnode[c.grp + 32 * k] * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(X) + off),
                                       (__attribute__((address_space(3))) void *)(Xh4 + (c.grp + 32 * (k + 1)) * PG::LPR + c.q), 16, 0, 16);
    }
  }
  wait_vmcnt0();
  __syncthreads();
}

// sum of the row's neighbours (slot bytes, in CSR order) + its own row (self loop), all from LDS
// (plain adds on purpose: this file is built without SLP packing, and written as v_pk_add_f32 -- f4_add_pk -- these sums cost the
// headline 2 %: they run beside the CU's other workgroup's MFMAs, where packed f32 VALU is slow; profiles/r05_x_ab_slp.txt)
// the row's 32 slot bytes, fetched from LDS BEFORE the wait (one address per 16-lane group: broadcast reads): they are live
// only across the wait and the gather, where registers are plentiful, and the aggregation does not start with a dependent read
__device__ __forceinline__ void tile_slot_words(const TileCtx &c, unsigned (&w)[8]) {
  const uint4 a = reinterpret_cast<const uint4 *>(c.lds_slots)[c.grp * 2], b = reinterpret_cast<const uint4 *>(c.lds_slots)[c.grp * 2 + 1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}

__device__ __forceinline__ float4 tile_aggregate(const TileCtx &c, const unsigned (&sw)[8], const float *ldsXh) {
  const float4 *Xh4 = reinterpret_cast<const float4 *>(ldsXh);
  float4 a = f4_zero();
#pragma unroll
  for (int jw = 0; jw < 8; ++jw) {
    if (jw * 4 < c.wmax) {   // wave-uniform
      const unsigned w = sw[jw];
      float4 v[4];
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) v[jb] = Xh4[((w >> (8 * jb)) & 0xff) * PG::LPR + c.q];
      a = f4_add(a, f4_add(f4_add(v[0], v[1]), f4_add(v[2], v[3])));
    }
  }
  return f4_add(a, Xh4[c.grp * PG::LPR + c.q]);
}
// the same sums in the same order with the slot words read from LDS round by round (8 fewer registers across the rounds; for the
// interleaved adjoint, which is at the 128-register edge)
__device__ __forceinline__ float4 tile_aggregate_lean(const TileCtx &c, const float *ldsXh) {
  const float4 *Xh4 = reinterpret_cast<const float4 *>(ldsXh);
  float4 a = f4_zero();
#pragma unroll 1
  for (int jw = 0; jw * 4 < c.wmax; ++jw) {   // wave-uniform
    const unsigned w = c.lds_slots[c.grp * 8 + jw];
    float4 v[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) v[jb] = Xh4[((w >> (8 * jb)) & 0xff) * PG::LPR + c.q];
    a = f4_add(a, f4_add(f4_add(v[0], v[1]), f4_add(v[2], v[3])));
  }
  return f4_add(a, Xh4[c.grp * PG::LPR + c.q]);
}

// weighted rows: the replayed plan's order of operations (halo_finish, gcn_fused.hip: one fma per slot, in slot order); the four
// weights of a round are one 16-byte LDS read that the row's 16 lanes share
__device__ __forceinline__ float4 tile_aggregate_weighted(const TileCtx &c, const float *ldsXh) {
  const float4 *Xh4 = reinterpret_cast<const float4 *>(ldsXh);
  float4 a = f4_zero();
#pragma unroll 1
  for (int jw = 0; jw * 4 < c.wmax; ++jw) {   // wave-uniform
    const unsigned w = c.lds_slots[c.grp * 8 + jw];
    const float4 wv = *reinterpret_cast<const float4 *>(c.lds_w + c.grp * kSlotWidth + 4 * jw);
    float4 v[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) v[jb] = Xh4[((w >> (8 * jb)) & 0xff) * PG::LPR + c.q];
    a = f4_fma(wv.x, v[0], a); a = f4_fma(wv.y, v[1], a); a = f4_fma(wv.z, v[2], a); a = f4_fma(wv.w, v[3], a);
  }
  return f4_add(a, Xh4[c.grp * PG::LPR + c.q]);
}
template <bool WGT>
__device__ __forceinline__ float4 tile_aggregate_rounds(const TileCtx &c, const float *ldsXh) {
  if constexpr (WGT) return tile_aggregate_weighted(c, ldsXh);
  else return tile_aggregate_lean(c, ldsXh);
}


// ---- own-first aggregation (round 6; the plan's OwnFirst tables, common.h) ------------------------------------------------------------
// The slot bytes a plan's kernels read list every row's own-tile slots first (padded to the wave's number of own rounds), then the
// foreign ones; TileMeta::of_pre names the own rounds per wave.  The one-tile kernels sum those rounds BEFORE they wait for their
// neighbours' flags -- a tile's own rows are in LDS since its last epilogue -- and only the foreign rounds behind the gather.  Measured
// with an in-kernel re-ordering of the same kind (profiles/r06_a_own_first.txt): forward launch 2.301 -> 2.234 ms.
// rounds [r0, r1) of the row's slot words from LDS, same association as tile_aggregate
__device__ __forceinline__ float4 tile_aggregate_rounds_range(const TileCtx &c, const unsigned (&sw)[8], const float *ldsXh, float4 a, int r0, int r1) {
  const float4 *Xh4 = reinterpret_cast<const float4 *>(ldsXh);
#pragma unroll
  for (int jw = 0; jw < 8; ++jw) {
    if (jw >= r0 && jw < r1) {   // wave-uniform
      const unsigned w = sw[jw];
      float4 v[4];
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) v[jb] = Xh4[((w >> (8 * jb)) & 0xff) * PG::LPR + c.q];
      a = f4_add(a, f4_add(f4_add(v[0], v[1]), f4_add(v[2], v[3])));
    }
  }
  return a;
}
// ---- hub geometry (graphs whose tiles do not fit the 96-row halo / 32-entry rows: BASELINE config 1's Cora-shaped graph) --------
// One workgroup per CU and tile; per tile and direction (lists built by node_persistent_setup, not part of the graph handle):
//   * a halo of up to kHubHalo = 256 distinct rows (own rows first), 64 KB of LDS -- the reach of an LDS-DMA destination;
//   * per row a VARIABLE-length list of slot bytes in CSR order (start aligned to 4, {start, length} per row), at most kHubList bytes
//     per tile; no zero row: the tail of a list is masked, not padded;
//   * rows longer than kSlotWidth entries ("long rows", the hubs) are summed by all 32 lane groups together: group g takes entries
//     g, g + 32, ..., the 32 partial rows meet in LDS and the row's own group adds them in group order;
//   * a wait list of up to 255 tiles: four flags per lane of the polling wave (position 63 of the list is the abort word's lane).
// Arithmetic per row otherwise as in the 96-row geometry (groups of four slots, then the own row).
constexpr int kHubHalo = 256;
constexpr int kHubList = 4096;
constexpr int kHubNbr = 256;
constexpr int kHubXhF = kHubHalo * PD;
constexpr int kHubMetaF = kHubList / 4 + (kHubHalo - kTM) + 2 * kTM + kTM / 4 + kHubList;   // slot bytes, foreign node ids, {start, length} per row, long-row indices, entry weights

struct HubCtx : TileCtx {
  const uint8_t *hs;         // LDS: the tile's slot bytes
  const int2 *hrows;         // LDS: {start, length} of every row's list
  const uint8_t *hlong;      // LDS: rows with more than kSlotWidth entries
  const float *hw;           // LDS: the entries' edge weights (same positions as the slot bytes), or NULL: unweighted graph
  int start, len;            // this row's list (len = 0 for a long row: it is summed cooperatively)
  int n_long;
  int nb1, nb2, nb3;         // wave 0: wait-list entries lane + 64, + 128, + 192 (my_nbr = entry lane)
};

__device__ __forceinline__ void hub_ctx_init(const TileMeta &m, HubCtx &c, float *lds_meta) {
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.wave_u = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  c.grp = c.tid >> 4;
  c.q = c.tid & 15;
  c.tile = xcd_tile(blockIdx.x, m.n_tiles);
  const size_t pos = (size_t)c.tile * kTM + c.grp;
  const int4 sc = m.hub_sched[pos];
  c.valid = sc.x >= 0;
  c.node = max(sc.x, 0);
  c.ci = c.valid ? __int_as_float(sc.w) : 0.f;
  const int4 info = m.hub_info[c.tile];   // {halo count, list bytes (multiple of 16), long rows, -}
  uint4 *ls = reinterpret_cast<uint4 *>(lds_meta);
  int *lh = reinterpret_cast<int *>(lds_meta + kHubList / 4);
  int2 *lr = reinterpret_cast<int2 *>(lds_meta + kHubList / 4 + (kHubHalo - kTM));
  unsigned *ll = reinterpret_cast<unsigned *>(lds_meta + kHubList / 4 + (kHubHalo - kTM) + 2 * kTM);
  if (c.tid * 16 < info.y) ls[c.tid] = reinterpret_cast<const uint4 *>(m.hub_slots + (size_t)c.tile * kHubList)[c.tid];
  if (c.tid < kHubHalo - kTM) lh[c.tid] = m.hub_halo[(size_t)c.tile * kHubHalo + kTM + c.tid];
  if (c.tid >= 256 && c.tid < 256 + kTM) lr[c.tid - 256] = m.hub_rows[(size_t)c.tile * kTM + (c.tid - 256)];
  if (c.tid >= 320 && c.tid < 320 + kTM / 4) ll[c.tid - 320] = reinterpret_cast<const unsigned *>(m.hub_long + (size_t)c.tile * kTM)[c.tid - 320];
  c.hs = reinterpret_cast<const uint8_t *>(ls);
  c.lds_hnode = lh;
  c.lds_slots = nullptr;
  c.lds_w = nullptr;
  c.hrows = lr;
  c.hlong = reinterpret_cast<const uint8_t *>(ll);
  c.hw = nullptr;
  if (m.hub_w) {   // (uniform) edge weights (src/layers.jl:206-231): one float beside every slot byte
    float4 *lw = reinterpret_cast<float4 *>(lds_meta + kHubList / 4 + (kHubHalo - kTM) + 2 * kTM + kTM / 4);
    const float4 *gw = reinterpret_cast<const float4 *>(m.hub_w + (size_t)c.tile * kHubList);
#pragma unroll
    for (int k = 0; k < kHubList / 4 / kThreads; ++k)
      if ((c.tid + k * kThreads) * 4 < info.y) lw[c.tid + k * kThreads] = gw[c.tid + k * kThreads];
    c.hw = reinterpret_cast<const float *>(lw);
  }
  c.hcount = __builtin_amdgcn_readfirstlane(info.x);
  c.n_long = __builtin_amdgcn_readfirstlane(info.z);
  const int2 mine = m.hub_rows[pos];
  c.start = mine.x;
  c.len = (c.valid && mine.y <= kSlotWidth) ? mine.y : 0;
  int wm = c.len;
  wm = max(wm, __shfl_xor(wm, 16));
  wm = max(wm, __shfl_xor(wm, 32));
  c.wmax = __builtin_amdgcn_readfirstlane(wm);
  const int *nb = m.nbr + (size_t)c.tile * kHubNbr;
  c.my_nbr = nb[c.lane]; c.nb1 = nb[c.lane + 64]; c.nb2 = nb[c.lane + 128]; c.nb3 = nb[c.lane + 192];
}

// tile_wait for a wait list of up to 255 tiles: four independent flag loads per lane and round
__device__ __forceinline__ bool hub_wait(const TileMeta &m, const HubCtx &c, int ph, int *s_ok) {
  if (ph <= 1) return true;
  if (c.wave_u == 0) {
    const unsigned need = (unsigned)(ph - 1);
    const unsigned *a0 = (c.lane == 63) ? m.abort_word : (c.my_nbr >= 0 ? m.flags + 32 * c.my_nbr : nullptr);
    const unsigned *a1 = c.nb1 >= 0 ? m.flags + 32 * c.nb1 : nullptr, *a2 = c.nb2 >= 0 ? m.flags + 32 * c.nb2 : nullptr,
                   *a3 = c.nb3 >= 0 ? m.flags + 32 * c.nb3 : nullptr;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    for (unsigned it = 1;; ++it) {
      unsigned f0 = need, f1 = need, f2 = need, f3 = need;
      if (a0) f0 = __hip_atomic_load(a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a1) f1 = __hip_atomic_load(a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a2) f2 = __hip_atomic_load(a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a3) f3 = __hip_atomic_load(a3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__any((int)(c.lane == 63 && f0 != 0))) { ok = false; break; }              // somebody gave up
      if (__all((int)((c.lane == 63 || f0 >= need) && f1 >= need && f2 >= need && f3 >= need))) break;
      if ((it & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
        if (c.lane == 0) __hip_atomic_store(m.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (c.lane == 0) *s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return *s_ok != 0;
}

// tile_gather_foreign for up to 224 foreign rows (halo slots 32 .. 255)
__device__ __forceinline__ void hub_gather_foreign(const HubCtx &c, const float *X, float *ldsXh) {
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
#pragma unroll
  for (int k = 0; k < (kHubHalo - kTM) / kTM; ++k) {
    if (4 * c.wave_u + 32 * (k + 1) < c.hcount) {   // wave-uniform: a wave's four groups stage four consecutive slots
      const unsigned off = (unsigned)c.lds_hnode[c.grp + 32 * k] * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(X) + off),
                                       (__attribute__((address_space(3))) void *)(Xh4 + (c.grp + 32 * (k + 1)) * PG::LPR + c.q), 16, 0, 16);
    }
  }
  wait_vmcnt0();
  __syncthreads();
}

// the row's first 32 slot bytes, fetched from LDS BEFORE the wait (as tile_slot_words; words beyond the row's list hold other rows'
// bytes or table words -- valid LDS, masked at use)
__device__ __forceinline__ void hub_slot_words(const HubCtx &c, unsigned (&w)[8]) {
  const unsigned *s4 = reinterpret_cast<const unsigned *>(c.hs + c.start);
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = s4[k];
}

// sum of the row's neighbours + its own row; `part` = 32 x 64 floats of LDS that nobody else uses during the aggregation.
// (Measured and not kept: the long row's partial rows formed first, eight masked entries per group unrolled with their slot bytes
// fetched together -- 8 + 8 LDS reads per lane whatever the row's length: the hub tile's aggregation 3.8 k -> 4.7 k cycles.)
// Weighted graphs (c.hw): one fma per entry in list order, as the 96-row geometry's weighted rows (tile_aggregate_weighted); a long
// row's groups fold weight * row into their partial sums.
__device__ __forceinline__ float4 hub_aggregate(const HubCtx &c, const unsigned (&sw)[8], const float *ldsXh, float *part) {
  const float4 *Xh4 = reinterpret_cast<const float4 *>(ldsXh);
  float4 a = f4_zero();
  if (c.hw) {   // (uniform)
#pragma unroll 1
    for (int jw = 0; jw * 4 < c.wmax; ++jw) {   // wave-uniform
      const unsigned w = sw[jw];
      const float4 wv = *reinterpret_cast<const float4 *>(c.hw + c.start + 4 * jw);   // (list starts are multiples of four: 16-byte aligned)
      const float ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const float4 v = f4_sel(4 * jw + jb < c.len, Xh4[((w >> (8 * jb)) & 0xff) * PG::LPR + c.q], f4_zero());
        a = f4_fma(4 * jw + jb < c.len ? ww[jb] : 0.f, v, a);
      }
    }
  } else {
#pragma unroll
    for (int jw = 0; jw < 8; ++jw) {
      if (jw * 4 < c.wmax) {   // wave-uniform
        const unsigned w = sw[jw];
        float4 v[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) v[jb] = f4_sel(4 * jw + jb < c.len, Xh4[((w >> (8 * jb)) & 0xff) * PG::LPR + c.q], f4_zero());
        a = f4_add_pk(a, f4_add_pk(f4_add_pk(v[0], v[1]), f4_add_pk(v[2], v[3])));
      }
    }
  }
  for (int li = 0; li < c.n_long; ++li) {   // uniform
    const int r = c.hlong[li];
    const int2 rl = c.hrows[r];
    float4 p = f4_zero();
    if (c.hw) {
      for (int j = c.grp; j < rl.y; j += kTM) p = f4_fma(c.hw[rl.x + j], Xh4[(unsigned)c.hs[rl.x + j] * PG::LPR + c.q], p);
    } else {
      for (int j = c.grp; j < rl.y; j += kTM) p = f4_add_pk(p, Xh4[(unsigned)c.hs[rl.x + j] * PG::LPR + c.q]);
    }
    reinterpret_cast<float4 *>(part)[c.grp * PG::LPR + c.q] = p;
    __syncthreads();
    if (c.wave_u == (r >> 2)) {   // the wave of the row's group: each of its four groups adds eight partial rows, two exchanges fold them
      const int g4 = c.grp & 3;
      float4 t = reinterpret_cast<const float4 *>(part)[g4 * PG::LPR + c.q];
#pragma unroll
      for (int k = 1; k < kTM / 4; ++k) t = f4_add_pk(t, reinterpret_cast<const float4 *>(part)[(g4 + 4 * k) * PG::LPR + c.q]);
      t.x += __shfl_xor(t.x, 16); t.y += __shfl_xor(t.y, 16); t.z += __shfl_xor(t.z, 16); t.w += __shfl_xor(t.w, 16);
      t.x += __shfl_xor(t.x, 32); t.y += __shfl_xor(t.y, 32); t.z += __shfl_xor(t.z, 32); t.w += __shfl_xor(t.w, 32);
      if (c.grp == r) a = f4_add_pk(a, t);
    }
    __syncthreads();   // (the buffer is written again: by the next long row, or by the caller -- the adjoint's dz tile)
  }
  return f4_add_pk(a, Xh4[c.grp * PG::LPR + c.q]);
}

// W (row-major [in][out]) -> LDS, transposed (forward: B[k = in][j = out], stored Bt[j][k]) or straight (pullback: Bt[j = in][k = out])
__device__ __forceinline__ void load_weight_lds(const float *wt, float *ldsBt, int tid, bool transpose) {
  if (transpose) {
    const int j = tid % PD, kg0 = tid / PD;
#pragma unroll
    for (int ps = 0; ps < PG::NPASS; ++ps) {
      const int kg = kg0 + ps * PG::KGP;
      if (kg < PD / 4) {
        const float *w = wt + (size_t)(4 * kg) * PD + j;
        *reinterpret_cast<float4 *>(&ldsBt[j * PG::TS + 4 * kg]) = make_float4(w[0], w[PD], w[2 * PD], w[3 * PD]);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < PG::W4; ++k) {
      const int idx = tid + k * kThreads;
      if (idx < PD * PD / 4) {
        const int wi = (idx * 4) / PD, wo = (idx * 4) % PD;
        *reinterpret_cast<float4 *>(&ldsBt[wi * PG::TS + wo]) = reinterpret_cast<const float4 *>(wt)[idx];
      }
    }
  }
}

// ---- pieces of the interleaved kernels' software pipeline -------------------------------------------------------------------
// wave 0: one flag load per lane of the wait list (lane 63: the abort word), NOT waited for
__device__ __forceinline__ unsigned poll_issue(const TileMeta &m, const TileCtx &c, const unsigned *flags) {
  const unsigned *addr = (c.lane == 63) ? m.abort_word : (c.my_nbr >= 0 ? flags + 32 * c.my_nbr : nullptr);
  unsigned f = (c.lane == 63) ? 0u : 0xffffffffu;
  if (addr) f = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return f;
}
// wave 0: did every tile of the wait list show phase ph - 1 (and nobody give up)?
__device__ __forceinline__ bool poll_ready(const TileCtx &c, unsigned f, int ph) {
  const unsigned need = (unsigned)(ph - 1);
  return __all((int)(c.lane == 63 ? f == 0u : f >= need)) != 0;
}
// a slot's own rows into halo slots 0..31 and the LDS-DMA of the foreign rows (tile_gather_foreign without its wait).  `halo`
// is __restrict__ so that LDS reads of OTHER regions issued behind it are not made to wait for the DMA (see dense_mfma.hip,
// products_beside_dma: behind a global_load_lds the wait-count pass otherwise drains vmcnt in front of every LDS access)
__device__ __forceinline__ void halo_fill_ahead(const TileCtx &c, const float *X, float *__restrict__ halo, float4 xown) {
  float4 *Xh4 = reinterpret_cast<float4 *>(halo);
  Xh4[c.grp * PG::LPR + c.q] = xown;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (4 * c.wave_u + 32 * (k + 1) < c.hcount) {   // wave-uniform
      const unsigned off = (unsigned)c.lds_hnode[c.grp + 32 * k] * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(X) + off),
                                       (__attribute__((address_space(3))) void *)(Xh4 + (c.grp + 32 * (k + 1)) * PG::LPR + c.q), 16, 0, 16);
    }
  }
}
// the same with the slot's OWN rows fetched as well (halo slots 0..31 = the tile's rows: one more DMA per wave; they were stored
// write-through at least a slot-phase earlier).  For kernels that keep no copy of them in registers.
__device__ __forceinline__ void halo_fill_all(const TileCtx &c, const float *X, float *__restrict__ halo) {
  float4 *Xh4 = reinterpret_cast<float4 *>(halo);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (k == 0 || 4 * c.wave_u + 32 * k < c.hcount) {   // wave-uniform
      const unsigned row = k == 0 ? (unsigned)c.node : (unsigned)c.lds_hnode[c.grp + 32 * (k - 1)];
      const unsigned off = row * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(X) + off),
                                       (__attribute__((address_space(3))) void *)(Xh4 + (c.grp + 32 * k) * PG::LPR + c.q), 16, 0, 16);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward solve
// ---------------------------------------------------------------------------------------------------------------------
struct PFwdK {
  TileMeta m;          // lists by TARGET
  int n_steps, S, act;
  int n_members;       // trajectories solved one after the other on the same structure (a block-diagonal batch of identical graphs)
  const float *u_in;   // [n_members][N][64]  c .* u0
  float *u_out;        // [n_members][N][64]  c .* u(T)
  float *bufA, *bufB;  // exchanged arrays: stage input (A), layer-1 output (B)
  const float *w1, *b1, *w2, *b2;
  float *tape;         // [n_members][n_steps][S][2][N][64] aggregated layer inputs, or null (forward-only plan)
  uint8_t *masks;      // [n_members][n_steps][S][2][mask_bytes] relu sign bits
  float *ztape;        // same shape as tape: the pre-activations, kept instead of the sign bits when the activation is not relu
  size_t row_elems, mask_bytes;
  size_t flag_stride;  // two-slot kernels: slot s uses bufA / bufB + s * row_elems and the flag words m.flags + s * flag_stride
  int pair_wgs;        // tile-pair mode of the two-slot kernels (PAIR): the grid; workgroup b holds tiles t and t + pair_wgs of ONE member
  int k_tiles;         // tile-round mode (node_fwd_persistentK_kernel): workgroup b holds tiles t, t + pair_wgs, ..., k_tiles of them
  float *state;        // ... and keeps their state in memory: [7][N][64] own rows -- u, k_0 .. k_5 (k rows zero at launch)
  const float *cf;     // device table [36 + 6]: cf[i * 6 + j], j < i: coefficient of k_j in the array written after stage i (next
                       // stage input / step update), 0 elsewhere; cf[36 + i]: coefficient of k_i itself.  Copied to LDS.
};

// WGT (edge weights, src/layers.jl:206-231): the 4 KB of slot weights of the tile need LDS that two resident W^T do not leave, so layer 1's
// W^T is B fragments in registers for the whole launch (16 per lane; the forward kernel has them to spare) and only W2^T is in LDS
// HUB: the hub geometry (256-row halo, variable-length slot lists, long rows shared by the 32 lane groups, one workgroup per CU)
template <int ACT, bool TAPE, bool WGT = false, bool HUB = false>
__global__ __launch_bounds__(kThreads, HUB ? 2 : 4) void node_fwd_persistent_kernel(const PFwdK p) {
  static_assert(!(HUB && WGT), "hub geometry: its own weight lists (HubCtx::hw), not the WGT form");
  constexpr int XH = HUB ? kHubXhF : kXhF, MF = HUB ? kHubMetaF : kMetaF;
  __shared__ __attribute__((aligned(16))) float lds[XH + 2 * kTileF + (WGT ? kWF + kSlotWF : 2 * kWF) + 2 * PD + MF + 48 + 4];
  static_assert(!HUB || sizeof(lds) <= 160 * 1024 - 64, "one workgroup per CU");
  float *ldsXh = lds, *ldsT = lds + XH, *ldsZ = ldsT + kTileF, *ldsW1 = ldsZ + kTileF, *ldsW2 = WGT ? ldsW1 : ldsW1 + kWF;
  float *ldsSW = ldsW2 + kWF, *ldsB = WGT ? ldsSW + kSlotWF : ldsSW;
  float *ldsMeta = ldsB + 2 * PD, *ldsC = ldsMeta + MF;
  int *s_ok = reinterpret_cast<int *>(ldsC + 48);
  typename std::conditional<HUB, HubCtx, TileCtx>::type c;
  if constexpr (HUB) hub_ctx_init(p.m, c, ldsMeta);
  else tile_ctx_init(p.m, c, ldsMeta);
  if (c.tid < 42) ldsC[c.tid] = p.cf[c.tid];
  const int act = ACT >= 0 ? ACT : p.act;
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
  float bw1[16];
  if constexpr (WGT) {
    reinterpret_cast<float2 *>(ldsSW)[c.tid] = reinterpret_cast<const float2 *>(p.m.slot_w + (size_t)c.tile * kSlotWF)[c.tid];   // [32][32]
    c.lds_w = ldsSW;
    const int i16 = c.lane & 15, kq = c.lane >> 4, ct = c.wave_u / PG::RT;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) bw1[4 * kb + r] = p.w1[(16 * kb + 4 * kq + r) * PD + 16 * ct + i16];
  } else {
    load_weight_lds(p.w1, ldsW1, c.tid, true);
  }
  load_weight_lds(p.w2, ldsW2, c.tid, true);
  if (c.tid < PD) ldsB[c.tid] = p.b1 ? p.b1[c.tid] : 0.f;
  else if (c.tid < 2 * PD) ldsB[c.tid] = p.b2 ? p.b2[c.tid - PD] : 0.f;
  if (!HUB && c.grp == 0) Xh4[kHaloCap * PG::LPR + c.q] = f4_zero();
  if (c.tid == 0) *s_ok = 1;
  const unsigned own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);   // byte offset of this thread's 16 bytes in a [N][64] array
  __syncthreads();
  // own rounds of this wave's rows (the plan's own-first tables; 0: everything is summed behind the gather, as without them)
  int of_pre = 0;
  if constexpr (!HUB && !WGT) {
    if (p.m.of_pre) of_pre = __builtin_amdgcn_readfirstlane((int)p.m.of_pre[(size_t)c.tile * 8 + c.wave_u]);
  }
  const float4 bias1 = reinterpret_cast<const float4 *>(ldsB)[c.q], bias2 = reinterpret_cast<const float4 *>(ldsB + PD)[c.q];
  bool ok = true;
  int ph = 0;   // phases count on across the members: a tile starts the next trajectory while its neighbours finish this one
  for (int mb = 0; mb < p.n_members; ++mb) {
  const float *u_in = p.u_in + (size_t)mb * p.row_elems;
  const size_t ev0 = (size_t)mb * p.n_steps * p.S * 2;
  float4 u = f4_sel(c.valid && ok, ld4_g(u_in, own), f4_zero());
  // six named values written through component-wise selects (f4_sel): an array written as `k[j] = (j == i) ? yv : k[j]` ends
  // up in scratch memory
  float4 k0 = f4_zero(), k1 = f4_zero(), k2 = f4_zero(), k3 = f4_zero(), k4 = f4_zero(), k5 = f4_zero();
  Xh4[c.grp * PG::LPR + c.q] = u;   // (nobody reads the halo slots between a publish and the next gather's barrier ...
  if constexpr (!HUB && !WGT) __syncthreads();   // ... except the own rounds of a member's first phase, summed ahead of any barrier)
  for (int n = 0; n < p.n_steps && ok; ++n) {
    for (int i = 0; i < p.S && ok; ++i) {
#pragma unroll
      for (int layer = 0; layer < 2; ++layer) {
        ++ph;
        const float *X = layer == 0 ? ((n == 0 && i == 0) ? u_in : p.bufA) : p.bufB;
        NGPDE_PST(p.m, ph, 0);
        float4 agg;
        if constexpr (HUB) {
          unsigned sw[8];
          hub_slot_words(c, sw);
          if (!hub_wait(p.m, c, ph, s_ok)) { ok = false; break; }
          NGPDE_PST(p.m, ph, 1);
          hub_gather_foreign(c, X, ldsXh);
          NGPDE_PST(p.m, ph, 2);
          agg = hub_aggregate(c, sw, ldsXh, ldsZ);   // (the product's output tile is free until this phase's product)
        } else if constexpr (!WGT) {
          unsigned sw[8];
          tile_slot_words(c, sw);
          float4 a = tile_aggregate_rounds_range(c, sw, ldsXh, f4_zero(), 0, of_pre);   // own rows: under the wait
          if (!tile_wait(p.m, c, ph, s_ok)) { ok = false; break; }
          NGPDE_PST(p.m, ph, 1);
          tile_gather_foreign(c, X, ldsXh);
          NGPDE_PST(p.m, ph, 2);
          a = tile_aggregate_rounds_range(c, sw, ldsXh, a, of_pre, (c.wmax + 3) >> 2);
          agg = f4_add(a, Xh4[c.grp * PG::LPR + c.q]);
        } else {
          unsigned sw[8];
          tile_slot_words(c, sw);
          if (!tile_wait(p.m, c, ph, s_ok)) { ok = false; break; }
          NGPDE_PST(p.m, ph, 1);
          tile_gather_foreign(c, X, ldsXh);
          NGPDE_PST(p.m, ph, 2);
          agg = tile_aggregate_weighted(c, ldsXh);
        }
        float4 acc = f4_scale(c.ci, agg);   // a_i = c_i * sum of the stored (pre-scaled) rows
        *reinterpret_cast<float4 *>(&ldsT[c.grp * PG::TS + 4 * c.q]) = acc;
        const size_t ev = ev0 + (size_t)(n * p.S + i) * 2 + layer;
        if (TAPE && c.valid) st4_stream_g(p.tape + ev * p.row_elems, own, acc);
        __syncthreads();
        NGPDE_PST(p.m, ph, 3);
        if (WGT && layer == 0) mfma_rows_times_bfrag64(ldsT, bw1, ldsZ, c.wave_u, c.lane);
        else mfma_rows_times_bt<PD>(ldsT, layer == 0 ? ldsW1 : ldsW2, ldsZ, c.wave_u, c.lane);
        __syncthreads();
        NGPDE_PST(p.m, ph, 4);
        const float4 z = f4_add(*reinterpret_cast<const float4 *>(&ldsZ[c.grp * PG::TS + 4 * c.q]), layer == 0 ? bias1 : bias2);
        const uint8_t sign_bits = (uint8_t)((z.x > 0.f ? 1 : 0) | (z.y > 0.f ? 2 : 0) | (z.z > 0.f ? 4 : 0) | (z.w > 0.f ? 8 : 0));
        float4 yv = f4_sel(c.valid, f4_scale(c.ci, f4_act(act, z)), f4_zero());   // stored as c .* y
        if (layer == 0) {
          if (c.valid) store_sc1(p.bufB, own, yv);
          Xh4[c.grp * PG::LPR + c.q] = yv;
        } else {
          // k_i = yv; next stage input (or the step update) = u + sum_j cf[i][j] k_j -- same order as the replayed plan:
          // coef_self * k_i first, then u, then k_0 .. k_{i-1}
          k0 = f4_sel(i == 0, yv, k0); k1 = f4_sel(i == 1, yv, k1); k2 = f4_sel(i == 2, yv, k2);
          k3 = f4_sel(i == 3, yv, k3); k4 = f4_sel(i == 4, yv, k4); k5 = f4_sel(i == 5, yv, k5);
          // (terms with a zero coefficient add an exact zero: k_j is finite, stale values of later stages included)
          float4 v = f4_scale(ldsC[36 + i], yv);
          v = f4_fma(1.0f, u, v);
          v = f4_fma(ldsC[i * 6 + 0], k0, v); v = f4_fma(ldsC[i * 6 + 1], k1, v); v = f4_fma(ldsC[i * 6 + 2], k2, v);
          v = f4_fma(ldsC[i * 6 + 3], k3, v); v = f4_fma(ldsC[i * 6 + 4], k4, v);
          if (i == p.S - 1) u = v;
          if (c.valid) store_sc1(p.bufA, own, v);
          Xh4[c.grp * PG::LPR + c.q] = v;
        }
        NGPDE_PST(p.m, ph, 5);
        tile_publish(p.m, c, ph);
        NGPDE_PST(p.m, ph, 6);
        // relu' for the adjoint (any other activation: the pre-activation itself): only the adjoint launch reads it, so it leaves
        // after the rows are published
        if constexpr (TAPE && ACT == NGPDE_ACT_RELU) stu8_g(p.masks + ev * p.mask_bytes + (size_t)c.tile * kThreads, (unsigned)c.tid, sign_bits);
        else if (TAPE && c.valid) st4_stream_g(p.ztape + ev * p.row_elems, own, z);
      }
    }
  }
  // (a tile writes its rows of u(T) only after all readers of its u0 rows are past that member's first phase)
  if (c.valid) st4_g(p.u_out + (size_t)mb * p.row_elems, own, f4_sel(ok, u, f4_nan()));
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// forward solve, K tiles per workgroup taking turns ("tile rounds"): graphs of more tiles than two per co-resident workgroup
// ---------------------------------------------------------------------------------------------------------------------
// Workgroup b holds tiles t, t + W, ..., t + (K - 1) W (W = the grid) and walks them in that order in EVERY phase.  Nothing of a
// tile lives in registers across its turns: u and the stage derivatives are own rows of p.state (the same thread writes and
// reads them), the slot / halo tables are re-read per turn, and ONE layer's W^T is in LDS at a time (staged once per phase for
// all K tiles).  No gather-ahead pipeline is needed: by the time a tile's turn comes again the workgroup has spent K - 1 turns on
// its other tiles, and the neighbours' rows and flags of the previous phase have long arrived (what the GAT solver's batch kernels
// showed, gat_fused.hip).  A workgroup's own tiles may even be neighbours: turn s of phase ph needs the others' phase ph - 1 only.
// Arithmetic per tile is node_fwd_persistent_kernel's, operation for operation.
template <int ACT, bool TAPE, bool WGT = false>
__global__ __launch_bounds__(kThreads, 4) void node_fwd_persistentK_kernel(const PFwdK p) {
  constexpr int kMS = meta_stride<WGT>(), kMT = meta_tiles<WGT>();
  __shared__ __attribute__((aligned(16))) float lds[kXhF + 2 * kTileF + kWF + 2 * PD + kMT * kMS + 48 + 4];
  static_assert(sizeof(lds) <= 80 * 1024 - 64, "two workgroups per CU");
  float *ldsXh = lds, *ldsT = lds + kXhF, *ldsZ = ldsT + kTileF, *ldsW = ldsZ + kTileF, *ldsB = ldsW + kWF;
  float *ldsMeta = ldsB + 2 * PD, *ldsC = ldsMeta + kMT * kMS;
  int *s_ok = reinterpret_cast<int *>(ldsC + 48);
  const int tid = threadIdx.x, q = tid & 15;
  if (tid < 42) ldsC[tid] = p.cf[tid];
  const int act = ACT >= 0 ? ACT : p.act;
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
  if (tid < PD) ldsB[tid] = p.b1 ? p.b1[tid] : 0.f;
  else if (tid < 2 * PD) ldsB[tid] = p.b2 ? p.b2[tid - PD] : 0.f;
  if (tid < PG::LPR) Xh4[kHaloCap * PG::LPR + tid] = f4_zero();
  if (tid == 0) *s_ok = 1;
  const int W = p.pair_wgs, K = min(p.k_tiles, kMT);
  const int t0 = xcd_tile(blockIdx.x, W);
  const unsigned rowb = (unsigned)(p.row_elems * sizeof(float));
  for (int s = 0; s < K && t0 + s * W < p.m.n_tiles; ++s) tile_tables_to_lds<WGT>(p.m, t0 + s * W, ldsMeta + s * kMS);
  __syncthreads();
  const float4 bias1 = reinterpret_cast<const float4 *>(ldsB)[q], bias2 = reinterpret_cast<const float4 *>(ldsB + PD)[q];
  bool ok = true;
  int ph = 0;
  for (int n = 0; n < p.n_steps && ok; ++n) {
    for (int i = 0; i < p.S && ok; ++i) {
#pragma unroll 1
      for (int layer = 0; layer < 2 && ok; ++layer) {
        ++ph;
        const float *X = layer == 0 ? ((n == 0 && i == 0) ? p.u_in : p.bufA) : p.bufB;
        const size_t ev = (size_t)(n * p.S + i) * 2 + layer;
        const bool last_phase = (n == p.n_steps - 1 && i == p.S - 1 && layer == 1);
        load_weight_lds(layer == 0 ? p.w1 : p.w2, ldsW, tid, true);   // (every wave is past the previous phase's products: its last publish)
        for (int s = 0; s < K; ++s) {
          const int tile = t0 + s * W;
          if (tile >= p.m.n_tiles) break;   // uniform
          TileCtx c;
          tile_ctx_from_lds<WGT>(c, tile, ldsMeta + s * kMS);
          const unsigned own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
          if (!tile_wait(p.m, c, ph, s_ok)) { ok = false; break; }
          halo_fill_all(c, X, ldsXh);
          // own rows of the state, in flight under the gather and the product: 0 = u, 1 + j = k_j (zero at launch, like the
          // registers of the one-tile kernel)
          float4 su = f4_zero(), sk0 = f4_zero(), sk1 = f4_zero(), sk2 = f4_zero(), sk3 = f4_zero(), sk4 = f4_zero();
          if (layer == 1) {
            su = (n == 0) ? f4_sel(c.valid, ld4_g(p.u_in, own), f4_zero()) : ld4_g(p.state, own);
            // (only the stage derivatives this stage combines, k_j with j < i: the others meet a zero coefficient)
            sk0 = i > 0 ? ld4_g(p.state, own + rowb) : f4_zero(); sk1 = i > 1 ? ld4_g(p.state, own + 2 * rowb) : f4_zero();
            sk2 = i > 2 ? ld4_g(p.state, own + 3 * rowb) : f4_zero(); sk3 = i > 3 ? ld4_g(p.state, own + 4 * rowb) : f4_zero();
            sk4 = i > 4 ? ld4_g(p.state, own + 5 * rowb) : f4_zero();
          }
          wait_vmcnt0();
          __syncthreads();   // halo rows landed (and the phase's W is in LDS)
          float4 acc = f4_scale(c.ci, tile_aggregate_rounds<WGT>(c, ldsXh));
          *reinterpret_cast<float4 *>(&ldsT[c.grp * PG::TS + 4 * c.q]) = acc;
          if (TAPE && c.valid) st4_stream_g(p.tape + ev * p.row_elems, own, acc);
          __syncthreads();
          mfma_rows_times_bt<PD>(ldsT, ldsW, ldsZ, c.wave_u, c.lane);
          __syncthreads();
          const float4 z = f4_add(*reinterpret_cast<const float4 *>(&ldsZ[c.grp * PG::TS + 4 * c.q]), layer == 0 ? bias1 : bias2);
          const uint8_t sign_bits = (uint8_t)((z.x > 0.f ? 1 : 0) | (z.y > 0.f ? 2 : 0) | (z.z > 0.f ? 4 : 0) | (z.w > 0.f ? 8 : 0));
          const float4 yv = f4_sel(c.valid, f4_scale(c.ci, f4_act(act, z)), f4_zero());
          if (layer == 0) {
            if (c.valid) store_sc1(p.bufB, own, yv);
          } else {
            const float4 k0 = i == 0 ? yv : sk0, k1 = i == 1 ? yv : sk1, k2 = i == 2 ? yv : sk2, k3 = i == 3 ? yv : sk3, k4 = i == 4 ? yv : sk4;   // k_i is yv itself
            float4 v = f4_scale(ldsC[36 + i], yv);
            v = f4_fma(1.0f, su, v);
            v = f4_fma(ldsC[i * 6 + 0], k0, v); v = f4_fma(ldsC[i * 6 + 1], k1, v); v = f4_fma(ldsC[i * 6 + 2], k2, v);
            v = f4_fma(ldsC[i * 6 + 3], k3, v); v = f4_fma(ldsC[i * 6 + 4], k4, v);
            if (c.valid) {
              if (i < p.S - 1) st4_g(p.state, own + (unsigned)(1 + i) * rowb, yv);   // (the last stage's derivative is combined here and never read again)
              if (i == p.S - 1) st4_g(p.state, own, v);
              if (last_phase) st4_g(p.u_out, own, v);
              store_sc1(p.bufA, own, v);
            }
          }
          tile_publish(p.m, c, ph);
          if constexpr (TAPE && ACT == NGPDE_ACT_RELU) stu8_g(p.masks + ev * p.mask_bytes + (size_t)c.tile * kThreads, (unsigned)c.tid, sign_bits);
          else if (TAPE && c.valid) st4_stream_g(p.ztape + ev * p.row_elems, own, z);
        }
      }
    }
  }
  if (!ok) {   // a wait was aborted: poison every row this workgroup owns
    __syncthreads();
    for (int s = 0; s < K; ++s) {
      const int tile = t0 + s * W;
      if (tile >= p.m.n_tiles) break;
      const int4 sc = p.m.sched[(size_t)tile * kTM + (tid >> 4)];
      if (sc.x >= 0) st4_g(p.u_out, (unsigned)sc.x * (unsigned)(PD * 4) + (unsigned)(q * 16), f4_nan());
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// forward solve, tile rounds with the NEXT turn's hand-off taken off the current turn's instruction stream
// ---------------------------------------------------------------------------------------------------------------------
// node_fwd_persistentK_kernel runs poll -> gather -> aggregate -> product -> epilogue -> drain -> flag one after the other for every
// turn (4.8 us), with only the CU's other workgroup to fill the three fabric round trips.  But a workgroup's NEXT turn depends on
// nothing this turn produces (turn s + 1 of a phase needs the neighbours' previous phase; turn 0 of the next phase needs what they
// finished K - 1 turns ago), so, as in the two-slot kernels (fwd_slot_phase):
//   T0  s_waitcnt vmcnt(0) + barrier: this turn's halo rows (gathered during the previous turn) have landed and the previous turn's
//       row stores are drained -> ITS flag goes out here (deferred publish); wave 0 issues the flag loads of the NEXT turn's wait
//       list (not waited for)
//   T1  LDS aggregation, operand tile, tape row; wave 0 looks at the flags; barrier (the halo region is free from here)
//   T2  if they were all there: the next turn's rows go out by LDS-DMA and travel under the product and the epilogue; product; barrier
//   T5  bias, activation, stage combination, row / state stores (not drained); then the next turn's state rows are fetched into the
//       registers this turn has just finished with
// A next turn whose flags were not there takes the blocking path at its T0.  Arithmetic per tile is the tile-round kernel's.
template <int ACT, bool TAPE>
__global__ __launch_bounds__(kThreads, 4) void node_fwd_persistentKP_kernel(const PFwdK p) {
  constexpr int kMS = meta_stride<false>(), kMT = meta_tiles<false>();
  __shared__ __attribute__((aligned(16))) float lds[kXhF + 2 * kTileF + kWF + 2 * PD + kMT * kMS + 48 + 4];
  static_assert(sizeof(lds) <= 80 * 1024 - 64, "two workgroups per CU");
  float *ldsXh = lds, *ldsT = lds + kXhF, *ldsZ = ldsT + kTileF, *ldsW = ldsZ + kTileF, *ldsB = ldsW + kWF;
  float *ldsMeta = ldsB + 2 * PD, *ldsC = ldsMeta + kMT * kMS;
  int *s_ok = reinterpret_cast<int *>(ldsC + 48), *s_pre = s_ok + 1;
  const int tid = threadIdx.x, q = tid & 15;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  if (tid < 42) ldsC[tid] = p.cf[tid];
  const int act = ACT >= 0 ? ACT : p.act;
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
  if (tid < PD) ldsB[tid] = p.b1 ? p.b1[tid] : 0.f;
  else if (tid < 2 * PD) ldsB[tid] = p.b2 ? p.b2[tid - PD] : 0.f;
  if (tid < PG::LPR) Xh4[kHaloCap * PG::LPR + tid] = f4_zero();
  if (tid == 0) *s_ok = 1, *s_pre = 0;
  const int W = p.pair_wgs, K = min(p.k_tiles, kMT);
  const int t0 = xcd_tile(blockIdx.x, W);
  const unsigned rowb = (unsigned)(p.row_elems * sizeof(float));
  int KT = 0;   // tiles this workgroup really holds (the last workgroups of a ragged grid hold one fewer)
  for (int s = 0; s < K && t0 + s * W < p.m.n_tiles; ++s, ++KT) tile_tables_to_lds<false>(p.m, t0 + s * W, ldsMeta + s * kMS);
  __syncthreads();
  const float4 bias1 = reinterpret_cast<const float4 *>(ldsB)[q], bias2 = reinterpret_cast<const float4 *>(ldsB + PD)[q];
  // no weight reload per phase (the tile-round kernel stages the phase's W^T in LDS every phase: ~1.2 k cycles per turn at four
  // tiles per workgroup): W2^T stays in LDS, W1^T is B fragments in registers for the whole launch (gcn_tile.h)
  float bw1[16];
  {
    const int i16 = lane & 15, kq = lane >> 4, ct = wave_u / PG::RT;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) bw1[4 * kb + r] = p.w1[(16 * kb + 4 * kq + r) * PD + 16 * ct + i16];
  }
  load_weight_lds(p.w2, ldsW, tid, true);
  bool ok = true, pre = false;
  unsigned *pend_flags = nullptr;
  int pend_ph = 0, n_ahead = 0;
  // this turn's state rows (layer-2 turns): fetched at the end of the previous turn when its gather went out ahead, else at T0
  float4 su = f4_zero(), sk0 = f4_zero(), sk1 = f4_zero(), sk2 = f4_zero(), sk3 = f4_zero(), sk4 = f4_zero();
  auto load_state = [&](const TileCtx &c, unsigned own, int n, int si) {   // si: the stage whose layer-2 turn reads them
    su = (n == 0) ? f4_sel(c.valid, ld4_g(p.u_in, own), f4_zero()) : ld4_g(p.state, own);
    // (only the stage derivatives that stage combines, k_j with j < si: the others meet a zero coefficient)
    sk0 = si > 0 ? ld4_g(p.state, own + rowb) : f4_zero(); sk1 = si > 1 ? ld4_g(p.state, own + 2 * rowb) : f4_zero();
    sk2 = si > 2 ? ld4_g(p.state, own + 3 * rowb) : f4_zero(); sk3 = si > 3 ? ld4_g(p.state, own + 4 * rowb) : f4_zero();
    sk4 = si > 4 ? ld4_g(p.state, own + 5 * rowb) : f4_zero();
  };
  int ph = 0;
  for (int n = 0; n < p.n_steps && ok; ++n) {
    for (int i = 0; i < p.S && ok; ++i) {
#pragma unroll 1
      for (int layer = 0; layer < 2 && ok; ++layer) {
        ++ph;
        const float *X = layer == 0 ? ((n == 0 && i == 0) ? p.u_in : p.bufA) : p.bufB;
        const size_t ev = (size_t)(n * p.S + i) * 2 + layer;
        const bool last_phase = (n == p.n_steps - 1 && i == p.S - 1 && layer == 1);
        for (int s = 0; s < KT; ++s) {
          const int tile = t0 + s * W;
          TileCtx c;
          tile_ctx_from_lds<false>(c, tile, ldsMeta + s * kMS);
          const unsigned own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
          // ---- T0
          [[maybe_unused]] const int turn = (ph - 1) * KT + s + 1;   // (stamps: one record per turn)
          NGPDE_PST(p.m, turn, 0);
          wait_vmcnt0();
          __syncthreads();
          NGPDE_PST(p.m, turn, 1);
          n_ahead += pre ? 1 : 0;
          if (pend_flags && tid == 0) __hip_atomic_store(pend_flags, (unsigned)pend_ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pend_flags = nullptr;
          if (!pre) {
            if (!tile_wait(p.m, c, ph, s_ok)) { ok = false; break; }
            halo_fill_all(c, X, ldsXh);
            if (layer == 1) load_state(c, own, n, i);
            wait_vmcnt0();
            __syncthreads();
          }
          NGPDE_PST(p.m, turn, 2);
          // ---- the turn after this one: the next tile of this phase, or this workgroup's first tile in the next phase
          const bool same_phase = s + 1 < KT;
          const int ns = same_phase ? s + 1 : 0, nph = same_phase ? ph : ph + 1, nlayer = same_phase ? layer : 1 - layer;
          const int nn = (same_phase || layer == 0) ? n : ((i == p.S - 1) ? n + 1 : n);
          const bool nexists = (same_phase || !last_phase) && KT > 1;   // (one tile: its own rows of the next phase are stored in THIS turn)
          const float *nX = same_phase ? X : (nlayer == 0 ? p.bufA : p.bufB);   // (only the launch's very first phase reads u_in)
          const float *nmeta = ldsMeta + ns * kMS;
          // its flags: fetched now, looked at behind the aggregation (they were published K - 1 turns ago: one look is enough)
          unsigned f1 = 0;
          if (nexists && wave_u == 0) {
            const int nb = reinterpret_cast<const int *>(nmeta + kMetaF + kTM * 4)[lane];
            const unsigned *addr = (lane == 63) ? p.m.abort_word : (nb >= 0 ? p.m.flags + 32 * nb : nullptr);
            f1 = (lane == 63) ? 0u : 0xffffffffu;
            if (addr) f1 = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          // ---- T1
          float4 acc = f4_scale(c.ci, tile_aggregate_lean(c, ldsXh));
          *reinterpret_cast<float4 *>(&ldsT[c.grp * PG::TS + 4 * c.q]) = acc;
          if (TAPE && c.valid) st4_stream_g(p.tape + ev * p.row_elems, own, acc);
          if (wave_u == 0) {
            const unsigned need = (unsigned)(nph - 1);
            const bool hit = nexists && __all((int)(lane == 63 ? f1 == 0u : f1 >= need)) != 0;
            if (lane == 0) *s_pre = hit ? 1 : 0;
          }
          __syncthreads();
          // ---- T2: the halo region is free; the next turn's rows travel under the product and the epilogue
          NGPDE_PST(p.m, turn, 3);
          pre = *s_pre != 0;
          TileCtx cn;
          unsigned ownn = 0;
          if (pre) {
            tile_ctx_from_lds<false>(cn, t0 + ns * W, nmeta);
            ownn = (unsigned)cn.node * (unsigned)(PD * 4) + (unsigned)(cn.q * 16);
            halo_fill_all(cn, nX, ldsXh);
          }
          NGPDE_PST(p.m, turn, 4);
          if (layer == 0) mfma_rows_times_bfrag64(ldsT, bw1, ldsZ, wave_u, lane);
          else mfma_rows_times_bt<PD>(ldsT, ldsW, ldsZ, wave_u, lane);
          __syncthreads();
          NGPDE_PST(p.m, turn, 5);
          // ---- T5
          const float4 z = f4_add(*reinterpret_cast<const float4 *>(&ldsZ[c.grp * PG::TS + 4 * c.q]), layer == 0 ? bias1 : bias2);
          const float cself = ldsC[36 + i], cf0 = ldsC[i * 6 + 0], cf1 = ldsC[i * 6 + 1], cf2 = ldsC[i * 6 + 2], cf3 = ldsC[i * 6 + 3],
                      cf4 = ldsC[i * 6 + 4];
          const uint8_t sign_bits = (uint8_t)((z.x > 0.f ? 1 : 0) | (z.y > 0.f ? 2 : 0) | (z.z > 0.f ? 4 : 0) | (z.w > 0.f ? 8 : 0));
          const float4 yv = f4_sel(c.valid, f4_scale(c.ci, f4_act(act, z)), f4_zero());
          if (layer == 0) {
            if (c.valid) store_sc1(p.bufB, own, yv);
          } else {
            const float4 k0 = i == 0 ? yv : sk0, k1 = i == 1 ? yv : sk1, k2 = i == 2 ? yv : sk2, k3 = i == 3 ? yv : sk3, k4 = i == 4 ? yv : sk4;
            float4 v = f4_scale(cself, yv);
            v = f4_fma(1.0f, su, v);
            v = f4_fma(cf0, k0, v); v = f4_fma(cf1, k1, v); v = f4_fma(cf2, k2, v);
            v = f4_fma(cf3, k3, v); v = f4_fma(cf4, k4, v);
            if (c.valid) {
              if (i < p.S - 1) st4_g(p.state, own + (unsigned)(1 + i) * rowb, yv);   // (the last stage's derivative is combined here and never read again)
              if (i == p.S - 1) st4_g(p.state, own, v);
              if (last_phase) st4_g(p.u_out, own, v);
              store_sc1(p.bufA, own, v);
            }
          }
          if (pre && nlayer == 1) load_state(cn, ownn, nn, i);   // (this turn's state rows are dead from here; a layer-2 turn that follows this one belongs to stage i)
          pend_flags = p.m.flags + 32 * tile;
          pend_ph = ph;
          if constexpr (TAPE && ACT == NGPDE_ACT_RELU) stu8_g(p.masks + ev * p.mask_bytes + (size_t)tile * kThreads, (unsigned)tid, sign_bits);
          else if (TAPE && c.valid) st4_stream_g(p.ztape + ev * p.row_elems, own, z);
          NGPDE_PST(p.m, turn, 6);
        }
      }
    }
  }
  wait_vmcnt0();
  __syncthreads();
  if (ok && pend_flags && tid == 0) __hip_atomic_store(pend_flags, (unsigned)pend_ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!ok) {   // a wait was aborted: poison every row this workgroup owns
    __syncthreads();
    for (int s = 0; s < KT; ++s) {
      const int4 sc = p.m.sched[(size_t)(t0 + s * W) * kTM + (tid >> 4)];
      if (sc.x >= 0) st4_g(p.u_out, (unsigned)sc.x * (unsigned)(PD * 4) + (unsigned)(q * 16), f4_nan());
    }
  }
  if (tid == 0 && p.m.stats) p.m.stats[2 * t0] = n_ahead;
}


// ---------------------------------------------------------------------------------------------------------------------
// forward solve, TWO trajectories of a batch interleaved in one workgroup
// ---------------------------------------------------------------------------------------------------------------------
// A block-diagonal batch of identical structures (test/runtests.jl:89-102, src/layers.jl:359-361) solved member after member
// pays the exposed hand-off (drain -> flag -> detect -> gather, ~2.4 us of a 4.1 us phase) once per member and phase: the CU idles
// while the rows travel.  Here a workgroup keeps the SAME tile of two members ("slots") and alternates between them phase by
// phase: while slot 0's rows and flag cross the fabric the workgroup aggregates, multiplies and stores slot 1's phase, and vice
// versa.  Each slot has its own exchanged arrays (bufA / bufB + slot * row_elems) and its own flag line per tile
// (flags + slot * flag_stride); wait lists, halo lists, slot bytes and W are shared.  The members are taken in pairs (0, 1),
// (2, 3), ...; an odd last member runs alone in slot 0.  Per-slot state in registers: u, k_0..k_5 and the slot's own row of the
// array it last published (the LDS halo region is shared by the slots, so the own rows are re-written at every phase start).
// Arithmetic per member is that of node_fwd_persistent_kernel operation for operation: u(T) is bitwise equal.
struct FSlot {
  float4 u, k0, k1, k2, k3, k4, k5, xown;
};

// What the fabric costs a slot-phase when its steps are simply run one after the other: the poll (a load that must reach memory),
// the gather of the foreign rows, the drain of the row stores in front of the flag -- three dependent round trips, ~2 us of a
// 3.8 us slot-phase, and alternating the slots alone hides none of them (measured: 18.3 ms for 8 members against 19.2 one by
// one).  So the round trips of the NEXT slot-phase are taken off the instruction stream of the current one:
//   T0  s_waitcnt vmcnt(0) + barrier: this slot-phase's halo rows have landed (gathered during the previous slot-phase) and the
//       previous slot-phase's row stores are drained -> ITS flag is published here (deferred publish: no wait of its own)
//   T1  LDS aggregation, operand tile, tape row
//   T2  barrier; wave 0 issues the flag loads of the NEXT slot-phase's wait list (not waited for)
//   T3  MFMA
//   T4  wave 0 looks at the flags it fetched; barrier
//   T5  if they were all there: the next slot-phase's own rows and the LDS-DMA of its foreign rows go out now (the halo region has
//       been free since T2) and fly during the epilogue; then bias, activation, stage combination, row stores (not drained)
// A next slot-phase whose flags were not there yet (or that does not exist: the last one, a single slot) takes the blocking
// path at its T0: publish what is pending, poll, gather -- the round-2 sequence.
struct FNext {          // the slot-phase after this one
  bool exists;
  int ph;               // its phase number
  const float *X;       // the array its halo rows come from
  const unsigned *flags;
};

// PAIR = false: the slots are the SAME tile of two members of a batch (c and cn are one context).  PAIR = true: the slots are two
// TILES of one member -- a graph of more tiles than co-resident workgroups (up to twice as many) -- with their own contexts; the
// exchanged arrays and the flag array are then common to both slots.
template <int ACT, bool TAPE, bool PAIR>
__device__ __forceinline__ bool fwd_slot_phase(const PFwdK &p, const TileCtx &c, const TileCtx &cn, FSlot &S, FSlot &Snext, const int sl, const int ph, const int n,
                                               const int i, const int layer, const float *X, const size_t ev0, const int act,
                                               const unsigned own, bool &pre, int &n_ahead, unsigned *&pend_flags, int &pend_ph, const FNext nx,
                                               float *ldsXh, float *ldsT, float *ldsZ, const float *ldsW, const float *ldsBias,
                                               const float *ldsC, int *s_ok, int *s_pre) {
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
  float *bufA = p.bufA + (PAIR ? 0 : (size_t)sl * p.row_elems), *bufB = p.bufB + (PAIR ? 0 : (size_t)sl * p.row_elems);
  unsigned *flags = p.m.flags + (PAIR ? 0 : (size_t)sl * p.flag_stride);
  NGPDE_PST(p.m, ph, 0);
  // T0
  wait_vmcnt0();
  __syncthreads();
  unsigned sw[8];
  tile_slot_words(c, sw);
  n_ahead += pre ? 1 : 0;
  if (pend_flags && c.tid == 0) __hip_atomic_store(pend_flags, (unsigned)pend_ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  pend_flags = nullptr;
  if (!pre) {
    Xh4[c.grp * PG::LPR + c.q] = S.xown;   // (behind the barrier: nobody is still aggregating from the halo region)
    if (!tile_wait(p.m, c, ph, s_ok, flags)) return false;
    NGPDE_PST(p.m, ph, 1);
    tile_gather_foreign(c, X, ldsXh);
  }
  NGPDE_PST(p.m, ph, 2);
  // T1
  float4 acc = f4_scale(c.ci, tile_aggregate(c, sw, ldsXh));
  *reinterpret_cast<float4 *>(&ldsT[c.grp * PG::TS + 4 * c.q]) = acc;
  const size_t ev = ev0 + (size_t)(n * p.S + i) * 2 + layer;
  if (TAPE && c.valid) st4_stream_g(p.tape + ev * p.row_elems, own, acc);
  // T2
  __syncthreads();
  unsigned f1 = 0;
  if (nx.exists && c.wave_u == 0) f1 = poll_issue(p.m, cn, nx.flags);
  NGPDE_PST(p.m, ph, 3);
  // T3
  mfma_rows_times_bt<PD>(ldsT, ldsW, ldsZ, c.wave_u, c.lane);
  // T4.  (A workgroup that runs AHEAD of its neighbours looks too early -- their flags of the phase before are published at the
  // top of their current slot-phase -- and takes the blocking path at its next T0, after which it is behind and finds them; a
  // second look behind the epilogue's arithmetic was measured: the barrier and the wait for the second load cost what the saved
  // blocking paths gave, 14.9 against 14.1 ms for 8 members.)
  if (c.wave_u == 0) {
    const bool hit = nx.exists && poll_ready(c, f1, nx.ph);
    if (c.lane == 0) *s_pre = hit ? 1 : 0;
  }
  __syncthreads();
  NGPDE_PST(p.m, ph, 4);
  // T5.  Every LDS read of the epilogue comes BEFORE a DMA is issued (see halo_fill_ahead)
  pre = *s_pre != 0;
  const float4 z = f4_add(*reinterpret_cast<const float4 *>(&ldsZ[c.grp * PG::TS + 4 * c.q]), reinterpret_cast<const float4 *>(ldsBias)[c.q]);
  const float cself = ldsC[36 + i], cf0 = ldsC[i * 6 + 0], cf1 = ldsC[i * 6 + 1], cf2 = ldsC[i * 6 + 2], cf3 = ldsC[i * 6 + 3],
              cf4 = ldsC[i * 6 + 4];
  if (pre) halo_fill_ahead(cn, nx.X, ldsXh, Snext.xown);
  const uint8_t sign_bits = (uint8_t)((z.x > 0.f ? 1 : 0) | (z.y > 0.f ? 2 : 0) | (z.z > 0.f ? 4 : 0) | (z.w > 0.f ? 8 : 0));
  const float4 yv = f4_sel(c.valid, f4_scale(c.ci, f4_act(act, z)), f4_zero());
  float4 v = f4_zero();
  if (layer != 0) {
    v = f4_scale(cself, yv);
    v = f4_fma(1.0f, S.u, v);
    // (k_i = yv enters with the zero weight cf[i][i], as in node_fwd_persistent_kernel, where it is assigned first)
    v = f4_fma(cf0, f4_sel(i == 0, yv, S.k0), v); v = f4_fma(cf1, f4_sel(i == 1, yv, S.k1), v); v = f4_fma(cf2, f4_sel(i == 2, yv, S.k2), v);
    v = f4_fma(cf3, f4_sel(i == 3, yv, S.k3), v); v = f4_fma(cf4, f4_sel(i == 4, yv, S.k4), v);
  }
  if (layer == 0) {
    if (c.valid) store_sc1(bufB, own, yv);
    S.xown = yv;
  } else {
    S.k0 = f4_sel(i == 0, yv, S.k0); S.k1 = f4_sel(i == 1, yv, S.k1); S.k2 = f4_sel(i == 2, yv, S.k2);
    S.k3 = f4_sel(i == 3, yv, S.k3); S.k4 = f4_sel(i == 4, yv, S.k4); S.k5 = f4_sel(i == 5, yv, S.k5);
    if (i == p.S - 1) S.u = v;
    if (c.valid) store_sc1(bufA, own, v);
    S.xown = v;
  }
  NGPDE_PST(p.m, ph, 5);
  pend_flags = flags + 32 * c.tile;     // published at the next T0 (or behind the loops)
  pend_ph = ph;
  if (TAPE) stu8_g(p.masks + ev * p.mask_bytes + (size_t)c.tile * kThreads, (unsigned)c.tid, sign_bits);
  NGPDE_PST(p.m, ph, 6);
  return true;
}

template <int ACT, bool TAPE, bool PAIR>
__global__ __launch_bounds__(kThreads, 4) void node_fwd_persistent2_kernel(const PFwdK p) {
  __shared__ __attribute__((aligned(16))) float lds[kXhF + 2 * kTileF + 2 * kWF + 2 * PD + (PAIR ? 2 : 1) * kMetaF + 48 + 4];
  float *ldsXh = lds, *ldsT = lds + kXhF, *ldsZ = ldsT + kTileF, *ldsW1 = ldsZ + kTileF, *ldsW2 = ldsW1 + kWF, *ldsB = ldsW2 + kWF;
  float *ldsMeta = ldsB + 2 * PD, *ldsC = ldsMeta + (PAIR ? 2 : 1) * kMetaF;
  int *s_ok = reinterpret_cast<int *>(ldsC + 48), *s_pre = s_ok + 1;
  TileCtx c, c1s;
  tile_ctx_init(p.m, c, ldsMeta, PAIR ? xcd_tile(blockIdx.x, p.pair_wgs) : -1);
  const bool has1 = !PAIR || c.tile + p.pair_wgs < p.m.n_tiles;   // (PAIR: an odd tile count leaves the last workgroups one tile)
  if (PAIR) tile_ctx_init(p.m, c1s, ldsMeta + kMetaF, has1 ? c.tile + p.pair_wgs : c.tile);
  const TileCtx &c1 = PAIR ? c1s : c;
  if (c.tid < 42) ldsC[c.tid] = p.cf[c.tid];
  const int act = ACT >= 0 ? ACT : p.act;
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
  load_weight_lds(p.w1, ldsW1, c.tid, true);
  load_weight_lds(p.w2, ldsW2, c.tid, true);
  if (c.tid < PD) ldsB[c.tid] = p.b1 ? p.b1[c.tid] : 0.f;
  else if (c.tid < 2 * PD) ldsB[c.tid] = p.b2 ? p.b2[c.tid - PD] : 0.f;
  if (c.grp == 0) Xh4[kHaloCap * PG::LPR + c.q] = f4_zero();
  if (c.tid == 0) *s_ok = 1, *s_pre = 0;
  const unsigned own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
  const unsigned own1 = (unsigned)c1.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
  __syncthreads();
  bool ok = true;
  int ph = 0;   // both slots run the same phase numbers; the count runs on across the pairs
  bool pre = false;
  int n_ahead = 0;
  unsigned *pend_flags = nullptr;
  int pend_ph = 0;
  const int NP = p.n_steps * p.S * 2;   // phases of one member
  for (int mb = 0; mb < p.n_members && ok; mb += 2) {
    const bool two = PAIR ? has1 : mb + 1 < p.n_members;
    const float *u_in0 = p.u_in + (size_t)mb * p.row_elems, *u_in1 = u_in0 + ((two && !PAIR) ? p.row_elems : 0);
    const size_t ev00 = (size_t)mb * NP, ev01 = PAIR ? ev00 : ev00 + NP;
    const size_t boff1 = PAIR ? 0 : p.row_elems;            // slot 1's offset in the exchanged arrays
    unsigned *flags1 = p.m.flags + (PAIR ? 0 : p.flag_stride);
    FSlot s0, s1;
    s0.u = f4_sel(c.valid, ld4_g(u_in0, own), f4_zero());
    s1.u = f4_sel(c1.valid && two, ld4_g(u_in1, own1), f4_zero());
    s0.k0 = s0.k1 = s0.k2 = s0.k3 = s0.k4 = s0.k5 = f4_zero();
    s1.k0 = s1.k1 = s1.k2 = s1.k3 = s1.k4 = s1.k5 = f4_zero();
    s0.xown = s0.u;
    s1.xown = s1.u;
    int n = 0, i = 0;
    for (int P = 0; P < NP && ok; P += 2) {   // P: layer-1 phase of stage evaluation (n, i); P + 1: its layer-2 phase
#pragma unroll
      for (int layer = 0; layer < 2; ++layer) {
        ++ph;
        const float *ldsW = layer == 0 ? ldsW1 : ldsW2, *ldsBias = layer == 0 ? ldsB : ldsB + PD;
        // the arrays the halo rows of this phase / of the next phase come from, per slot
        const float *X0 = layer == 0 ? (P == 0 ? u_in0 : p.bufA) : p.bufB;
        const float *X1 = layer == 0 ? (P == 0 ? u_in1 : p.bufA + boff1) : p.bufB + boff1;
        FNext nx0, nx1;   // after (phase, slot 0): (phase, slot 1) if there are two slots; after (phase, slot 1): (phase + 1, slot 0)
        nx0.exists = two; nx0.ph = ph; nx0.X = X1; nx0.flags = flags1;
        nx1.exists = P + layer + 1 < NP; nx1.ph = ph + 1; nx1.X = layer == 0 ? p.bufB : p.bufA; nx1.flags = p.m.flags;
        if (!fwd_slot_phase<ACT, TAPE, PAIR>(p, c, c1, s0, s1, 0, ph, n, i, layer, X0, ev00, act, own, pre, n_ahead, pend_flags, pend_ph, nx0, ldsXh,
                                             ldsT, ldsZ, ldsW, ldsBias, ldsC, s_ok, s_pre)) { ok = false; break; }
        if (two && !fwd_slot_phase<ACT, TAPE, PAIR>(p, c1, c, s1, s0, 1, ph, n, i, layer, X1, ev01, act, own1, pre, n_ahead, pend_flags, pend_ph, nx1,
                                                    ldsXh, ldsT, ldsZ, ldsW, ldsBias, ldsC, s_ok, s_pre)) { ok = false; break; }
      }
      if (++i == p.S) i = 0, ++n;
    }
    // the last slot-phase's flag (the next pair's first phases wait for it; nothing was gathered ahead: pre is false here)
    wait_vmcnt0();
    __syncthreads();
    if (ok && pend_flags && c.tid == 0) __hip_atomic_store(pend_flags, (unsigned)pend_ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pend_flags = nullptr;
    // (a tile writes its rows of u(T) only after all readers of its u0 rows are past that member's first phase)
    if (c.valid) st4_g(p.u_out + (size_t)mb * p.row_elems, own, f4_sel(ok, s0.u, f4_nan()));
    if (two && c1.valid) st4_g(p.u_out + (PAIR ? 0 : (size_t)(mb + 1) * p.row_elems), own1, f4_sel(ok, s1.u, f4_nan()));
    if (PAIR) break;   // one member
  }
  if (!ok) {   // an aborted solve poisons every member's output
    for (int mb = 0; mb < p.n_members; ++mb) {
      if (c.valid) st4_g(p.u_out + (size_t)mb * p.row_elems, own, f4_nan());
      if (PAIR && has1 && c1.valid) st4_g(p.u_out + (size_t)mb * p.row_elems, own1, f4_nan());
    }
  }
  if (c.tid == 0 && p.m.stats) p.m.stats[2 * c.tile] = n_ahead;
}

// ---------------------------------------------------------------------------------------------------------------------
// discrete adjoint
// ---------------------------------------------------------------------------------------------------------------------
struct PBwdK {
  TileMeta m;          // lists by SOURCE
  int n_steps, S, n_members, act;
  const float *ztape;  // pre-activations (activations other than relu), or null
  float *lam;          // [n_members][N][64] in: dL/du~(T) (adjoint seed ./ c); out: dL/du~0
  float *g1, *g2;      // exchanged arrays: c .* (dZ1 W1^T), c .* (dZ2 W2^T)
  const float *w1, *w2;
  const float *tape;
  const uint8_t *masks;
  size_t row_elems, mask_bytes;
  float *slab_dw1, *slab_db1, *slab_dw2, *slab_db2;   // [n_tiles][...] written ONCE, at the end
  int pair_wgs;        // tile-pair mode (PAIR): the grid; workgroup b holds tiles t and t + pair_wgs of ONE member
  int k_tiles;         // tile-round mode (node_bwd_persistentK_kernel): k_tiles tiles per workgroup, state in lam / ubar (zero at launch)
  size_t flag_stride;  // two-slot kernel: slot s uses g1 / g2 + s * row_elems, the flag words m.flags + s * flag_stride and
  float *ubar;         // the stage-adjoint scratch ubar + s * 5 * row_elems ([slot][5][N][64])
  const float *cb;     // device table [6 + 36 + 6], copied to LDS: cb[j] = dt * b[j]; cb[6 + i * 6 + j], j > i >= 1: dt * a[j][i-1], the
                       // weight of U-bar_j in K-bar_{i-1}, 0 elsewhere; cb[42 + i] = dt * a[i][i-1], the weight of U-bar_i itself
};

// ACT = NGPDE_ACT_RELU: relu' from the forward launch's sign bits; ACT = -1: any activation (p.act), act'(z) from the saved
// pre-activations (one more tape row per phase)
// WGT: the tile's slot weights need 4 KB of LDS that two padded W do not leave (and the adjoint has no 16 registers to park
// fragments in: fetched per phase they spill).  W1 is kept UNPADDED and XOR-swizzled instead (16 KB, conflict-free fragment reads:
// gcn_tile.h, mfma_rows_times_bswz64), which makes exactly the room.
template <int ACT, bool WGT = false, bool HUB = false>
__global__ __launch_bounds__(kThreads, HUB ? 2 : 4) void node_bwd_persistent_kernel(const PBwdK p) {
  static_assert(!(HUB && WGT), "hub geometry: its own weight lists (HubCtx::hw), not the WGT form");
  constexpr bool RELU = (ACT == NGPDE_ACT_RELU);
  using Aux = typename std::conditional<RELU, unsigned, float4>::type;   // what act' is formed from: 4 sign bits / the row of z
  constexpr int XH = HUB ? kHubXhF : kXhF, MF = HUB ? kHubMetaF : kMetaF;
  __shared__ __attribute__((aligned(16))) float lds[XH + 2 * kTileF + (WGT ? PD * PD + kWF + kSlotWF : 2 * kWF) + MF + 48 + 4];
  static_assert(sizeof(lds) <= (HUB ? 160 : 80) * 1024 - 64, "two workgroups per CU (hub geometry: one)");
  float *ldsXh = lds, *ldsG = lds, *ldsDZ = lds + XH, *ldsX = ldsDZ + kTileF, *ldsW1 = ldsX + kTileF, *ldsW2 = ldsW1 + (WGT ? PD * PD : kWF);
  float *ldsSW = ldsW2 + kWF;
  float *ldsMeta = WGT ? ldsSW + kSlotWF : ldsSW, *ldsC = ldsMeta + MF;
  int *s_ok = reinterpret_cast<int *>(ldsC + 48);
  typename std::conditional<HUB, HubCtx, TileCtx>::type c;
  if constexpr (HUB) hub_ctx_init(p.m, c, ldsMeta);
  else tile_ctx_init(p.m, c, ldsMeta);
  if (c.tid < 48) ldsC[c.tid] = p.cb[c.tid];
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
  if constexpr (WGT) {
    reinterpret_cast<float2 *>(ldsSW)[c.tid] = reinterpret_cast<const float2 *>(p.m.slot_w + (size_t)c.tile * kSlotWF)[c.tid];
    c.lds_w = ldsSW;
#pragma unroll
    for (int k = 0; k < PG::W4; ++k) {   // W1 straight (B^T[j = in][k = out] = W1[j][k]), quad k4 of row j at quad k4 ^ (j & 15)
      const int idx = c.tid + k * kThreads;
      if (idx < PD * PD / 4) {
        const int wi = (idx * 4) / PD, w4 = ((idx * 4) % PD) / 4;
        *reinterpret_cast<float4 *>(&ldsW1[wi * PD + 4 * (w4 ^ (wi & 15))]) = reinterpret_cast<const float4 *>(p.w1)[idx];
      }
    }
  } else {
    load_weight_lds(p.w1, ldsW1, c.tid, false);
  }
  load_weight_lds(p.w2, ldsW2, c.tid, false);
  if (!HUB && c.grp == 0) Xh4[kHaloCap * PG::LPR + c.q] = f4_zero();
  if (c.tid == 0) *s_ok = 1;
  const unsigned own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
  constexpr int NT = PG::CT * PG::CT;
  f32x4 dw1[PG::DWT], dw2[PG::DWT];
#pragma unroll
  for (int mm = 0; mm < PG::DWT; ++mm) dw1[mm] = dw2[mm] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float db1 = 0.f, db2 = 0.f;
  const int dbc = c.tid / PG::DBP, dbpart = c.tid % PG::DBP;
  const int S = p.S;
  int of_pre = 0;   // own rounds of this wave's rows (the plan's own-first tables): summed before the wait
  if constexpr (!HUB && !WGT) {
    if (p.m.of_pre) of_pre = __builtin_amdgcn_readfirstlane((int)p.m.of_pre[(size_t)c.tile * 8 + c.wave_u]);
  }
  __syncthreads();

  // the dense half of a phase: dL/dy = c .* K-bar, relu' by the sign bits, G = dZ W^T -> c .* G stored for the next gather,
  // dW += A^T dZ, db += column sums; then publish and keep the own rows of G in the halo slots
  auto tape_row = [&](size_t ev) { return ld4_stream_g(p.tape + ev * p.row_elems, own); };
  auto mask_of = [&](size_t ev) -> Aux {
    if constexpr (RELU) return ldu8_g(p.masks + ev * p.mask_bytes + (size_t)c.tile * kThreads, (unsigned)c.tid);
    else return ld4_stream_g(p.ztape + ev * p.row_elems, own);
  };
  // The tape row and the sign bits of the NEXT phase (mk_n / xrow_n) are asked for between the publish and the parameter-gradient
  // products: cold rows from HBM, ~4.6 k cycles away.  Asked for at the head of their own phase -- in front of the poll, as rounds
  // 2 - 3 had it -- they hold the poll up, because a wave's loads return in order: the first look at the flags came back only behind
  // the tape row (wait 4.6 k cycles against the forward kernel's 3.0 k for the same hand-off).
  Aux mk_n{};
  float4 xrow_n = f4_zero();
  // The first look at the NEXT phase's flags is asked for right behind the publish, in front of the tape prefetch and the
  // parameter-gradient products (poll_issue), and read when the next phase begins (tile_wait_primed): a tile whose neighbours have all
  // published before it no longer pays a poll round trip at the top of the phase.  Measured (round 6, same box, alternating): adjoint
  // launch 2.81 -> 2.78 ms.  (Also measured there and NOT kept: wave 0's tape rows fetched memory -> LDS by DMA a phase early, or by
  // four helper waves, so that nothing of wave 0's is in flight in front of its polls -- the hypothesis being that a wave's loads return
  // in order and the cold tape lines hold the polls up: adjoint 2.93 - 2.95 ms, the wait unchanged at 2.6 - 2.8 k cycles; and every
  // added register spills here -- 123 of 128 are taken -- a spilled dW accumulator is reloaded with s_waitcnt vmcnt(0) in front of the
  // products, which then waits for the tape rows.)
  unsigned f_next = 0;
  auto dense = [&](int ph, const float *ldsW, f32x4 (&dwl)[PG::DWT], float &dbl, float4 kbar, Aux mk, float4 xrow, float *gout, bool pf, size_t ev_n) {
    kbar = f4_scale(c.ci, kbar);
    float4 dz;
    if constexpr (RELU) {
      dz = c.valid ? make_float4((mk & 1u) ? kbar.x : 0.f, (mk & 2u) ? kbar.y : 0.f, (mk & 4u) ? kbar.z : 0.f, (mk & 8u) ? kbar.w : 0.f)
                   : f4_zero();
    } else {
      dz = f4_sel(c.valid, f4_mul(kbar, f4_dact(p.act, mk)), f4_zero());
    }
    *reinterpret_cast<float4 *>(&ldsDZ[c.grp * PG::TS + 4 * c.q]) = dz;
    *reinterpret_cast<float4 *>(&ldsX[c.grp * PG::TS + 4 * c.q]) = f4_sel(c.valid, xrow, f4_zero());
    __syncthreads();
    NGPDE_PST(p.m, ph, 3);
    if (WGT && ldsW == ldsW1) mfma_rows_times_bswz64(ldsDZ, ldsW, ldsG, c.wave_u, c.lane);   // (layer 1 of a weighted graph: the unpadded, swizzled W1)
    else mfma_rows_times_bt<PD>(ldsDZ, ldsW, ldsG, c.wave_u, c.lane);
    __syncthreads();
    NGPDE_PST(p.m, ph, 4);
    const float4 gv = f4_sel(c.valid, f4_scale(c.ci, *reinterpret_cast<const float4 *>(&ldsG[c.grp * PG::TS + 4 * c.q])), f4_zero());
    if (c.valid) store_sc1(gout, own, gv);
    NGPDE_PST(p.m, ph, 5);
    tile_publish(p.m, c, ph);
    NGPDE_PST(p.m, ph, 6);
    Xh4[c.grp * PG::LPR + c.q] = gv;   // behind the barrier: every thread has read its row of G (same LDS region)
    if constexpr (!HUB) {
      if (c.wave_u == 0) f_next = poll_issue(p.m, c, p.m.flags);
    }
    if (pf) {
      mk_n = mask_of(ev_n);
      xrow_n = tape_row(ev_n);
    }
    // The parameter-gradient products run AFTER the rows are published: nobody waits for them, so they fill the time the
    // neighbours need to see the flag and this tile needs to see theirs.  (The operand tiles stay intact until the next
    // phase's dense half, two barriers away.)
    // dWt[i][o] += sum_n A[n][i] dZ[n][o] over the tile's 32 rows
    const int i16 = c.lane & 15, kq = c.lane >> 4;
#pragma unroll
    for (int mm = 0; mm < PG::DWT; ++mm) {
      const int tt = c.wave_u + PG::WAVES * mm;
      if (tt < NT) {   // wave-uniform
        const int mt = tt / PG::CT, nt = tt % PG::CT;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {   // two halves of the 32-row contraction: 8 operand registers live instead of 16
          float a[kTM / 8], b[kTM / 8];
#pragma unroll
          for (int ks = 0; ks < kTM / 8; ++ks) {
            a[ks] = ldsX[(4 * (ks + 4 * kh) + kq) * PG::TS + mt * 16 + i16];
            b[ks] = ldsDZ[(4 * (ks + 4 * kh) + kq) * PG::TS + nt * 16 + i16];
          }
#pragma unroll
          for (int ks = 0; ks < kTM / 8; ++ks) dwl[mm] = mfma16(a[ks], b[ks], dwl[mm]);
        }
      }
    }
    {
      float s = 0.f;
#pragma unroll
      for (int nn = dbpart; nn < kTM; nn += PG::DBP) s += ldsDZ[nn * PG::TS + dbc];
#pragma unroll
      for (int o = 1; o < PG::DBP; o <<= 1) s += __shfl_xor(s, o);
      dbl += s;
    }
    NGPDE_PST(p.m, ph, 7);
  };

  bool ok = true;
  int ph = 0;   // phases count on across the members (the parameter-gradient accumulators too: the gradient of a batch is the sum)
  for (int mb = 0; mb < p.n_members; ++mb) {
  float *lam_g = p.lam + (size_t)mb * p.row_elems;
  const size_t ev0 = (size_t)mb * p.n_steps * S * 2;
  float4 lam = f4_sel(c.valid && ok, ld4_g(lam_g, own), f4_zero());
  float4 ub1 = f4_zero(), ub2 = f4_zero(), ub3 = f4_zero(), ub4 = f4_zero(), ub5 = f4_zero();   // named, not an array (see the forward kernel)
  if (ok) {   // first phase of a member: K-bar of the last stage of the last step = dt b_S lambda, layer 2's dense half (no gather:
              // g2 was last read two phases ago, so no wait either)
    ++ph;
    const size_t ev = ev0 + (size_t)((p.n_steps - 1) * S + (S - 1)) * 2 + 1;
    dense(ph, ldsW2, dw2, db2, f4_scale(ldsC[S - 1], lam), mask_of(ev), tape_row(ev), p.g2, true,
          ev0 + (size_t)((p.n_steps - 1) * S + (S - 1)) * 2);   // (next: layer 1 of the last stage of the last step)
  }
  for (int n = p.n_steps - 1; n >= 0 && ok; --n) {
    for (int i = S - 1; i >= 0 && ok; --i) {
      {   // layer 1 of stage i: dL/dy1 = A^T g2
        ++ph;
        NGPDE_PST(p.m, ph, 0);
        const Aux mk = mk_n;                      // asked for behind the publish of the phase before
        const float4 xrow = xrow_n;
        // next: layer 2 of the stage evaluated before this one (none in the very last phase of the member)
        const bool last_next = (i == 0 && n == 0);
        const size_t ev_next = ev0 + ((i >= 1) ? (size_t)(n * S + i - 1) * 2 + 1 : (size_t)((max(n, 1) - 1) * S + (S - 1)) * 2 + 1);
        float4 t;
        if constexpr (HUB) {
          unsigned sw[8];
          hub_slot_words(c, sw);
          if (!hub_wait(p.m, c, ph, s_ok)) { ok = false; break; }
          NGPDE_PST(p.m, ph, 1);
          hub_gather_foreign(c, p.g2, ldsXh);
          NGPDE_PST(p.m, ph, 2);
          t = hub_aggregate(c, sw, ldsXh, ldsDZ);   // (the operand tiles of the previous phase's products are dead: that phase ended at this wait's barrier)
        } else if constexpr (WGT) {
          if (!tile_wait_primed(p.m, c, ph, s_ok, p.m.flags, f_next)) { ok = false; break; }
          NGPDE_PST(p.m, ph, 1);
          tile_gather_foreign(c, p.g2, ldsXh);
          NGPDE_PST(p.m, ph, 2);
          t = tile_aggregate_weighted(c, ldsXh);
        } else {
          unsigned sw[8];
          tile_slot_words(c, sw);
          float4 a = tile_aggregate_rounds_range(c, sw, ldsXh, f4_zero(), 0, of_pre);   // own rows (in LDS since the last publish): under the wait
          if (!tile_wait_primed(p.m, c, ph, s_ok, p.m.flags, f_next)) { ok = false; break; }
          NGPDE_PST(p.m, ph, 1);
          tile_gather_foreign(c, p.g2, ldsXh);
          NGPDE_PST(p.m, ph, 2);
          a = tile_aggregate_rounds_range(c, sw, ldsXh, a, of_pre, (c.wmax + 3) >> 2);
          t = f4_add(a, Xh4[c.grp * PG::LPR + c.q]);
        }
        dense(ph, ldsW1, dw1, db1, t, mk, xrow, p.g1, !last_next, ev_next);
      }
      {   // U-bar_i = A^T g1; K-bar of the stage evaluated before it (or the lambda update), layer 2's dense half
        ++ph;
        const bool last = (i == 0 && n == 0);
        NGPDE_PST(p.m, ph, 0);
        const Aux mk = mk_n;                      // (the last phase of a member runs no dense half: nothing was asked for)
        const float4 xrow = xrow_n;
        // next: layer 1 of stage i - 1, or of the last stage of the step before
        const size_t ev_next = ev0 + (size_t)(i >= 1 ? n * S + i - 1 : (max(n, 1) - 1) * S + (S - 1)) * 2;
        float4 t;
        if constexpr (HUB) {
          unsigned sw[8];
          hub_slot_words(c, sw);
          if (!hub_wait(p.m, c, ph, s_ok)) { ok = false; break; }
          NGPDE_PST(p.m, ph, 1);
          hub_gather_foreign(c, p.g1, ldsXh);
          NGPDE_PST(p.m, ph, 2);
          t = hub_aggregate(c, sw, ldsXh, ldsDZ);   // (the operand tiles of the previous phase's products are dead: that phase ended at this wait's barrier)
        } else if constexpr (WGT) {
          if (!tile_wait_primed(p.m, c, ph, s_ok, p.m.flags, f_next)) { ok = false; break; }
          NGPDE_PST(p.m, ph, 1);
          tile_gather_foreign(c, p.g1, ldsXh);
          NGPDE_PST(p.m, ph, 2);
          t = tile_aggregate_weighted(c, ldsXh);
        } else {
          unsigned sw[8];
          tile_slot_words(c, sw);
          float4 a = tile_aggregate_rounds_range(c, sw, ldsXh, f4_zero(), 0, of_pre);   // own rows (in LDS since the last publish): under the wait
          if (!tile_wait_primed(p.m, c, ph, s_ok, p.m.flags, f_next)) { ok = false; break; }
          NGPDE_PST(p.m, ph, 1);
          tile_gather_foreign(c, p.g1, ldsXh);
          NGPDE_PST(p.m, ph, 2);
          a = tile_aggregate_rounds_range(c, sw, ldsXh, a, of_pre, (c.wmax + 3) >> 2);
          t = f4_add(a, Xh4[c.grp * PG::LPR + c.q]);
        }
        float4 kbar;
        if (i >= 1) {
          // same order as the replayed plan: coef_self * t, then lambda, then U-bar_{i+1} ..
          ub1 = f4_sel(i == 1, t, ub1); ub2 = f4_sel(i == 2, t, ub2); ub3 = f4_sel(i == 3, t, ub3);
          ub4 = f4_sel(i == 4, t, ub4); ub5 = f4_sel(i == 5, t, ub5);
          float4 v = f4_scale(ldsC[42 + i], t);
          v = f4_fma(ldsC[i - 1], lam, v);
          v = f4_fma(ldsC[6 + i * 6 + 2], ub2, v); v = f4_fma(ldsC[6 + i * 6 + 3], ub3, v);   // zero weights add an exact zero
          v = f4_fma(ldsC[6 + i * 6 + 4], ub4, v); v = f4_fma(ldsC[6 + i * 6 + 5], ub5, v);
          kbar = v;
        } else {
          float4 v = f4_scale(1.0f, t);
          v = f4_fma(1.0f, lam, v);
          v = f4_fma(1.0f, ub1, v); v = f4_fma(1.0f, ub2, v); v = f4_fma(1.0f, ub3, v);   // stage adjoints beyond S stay zero
          v = f4_fma(1.0f, ub4, v); v = f4_fma(1.0f, ub5, v);
          lam = v;
          kbar = f4_scale(ldsC[S - 1], v);
        }
        if (last) break;   // (this phase writes nothing other tiles read: no flag; the next member's first phase publishes ph + 1)
        dense(ph, ldsW2, dw2, db2, kbar, mk, xrow, p.g2, true, ev_next);
      }
    }
  }
  if (c.valid) st4_g(lam_g, own, f4_sel(ok, lam, f4_nan()));
  }
  // the tile's contribution to the parameter gradients: one slab per tile, summed by reduce_slabs_kernel
  const float bad = __int_as_float(0x7fc00000);
  auto write_slab = [&](const f32x4 (&dwl)[PG::DWT], float dbl, float *slab_dw, float *slab_db) {
    float4 *slab4 = reinterpret_cast<float4 *>(slab_dw + (size_t)blockIdx.x * PD * PD);
#pragma unroll
    for (int mm = 0; mm < PG::DWT; ++mm) {
      const int tt = c.wave_u + PG::WAVES * mm;
      if (tt < NT) slab4[tt * 64 + c.lane] = f4_sel(ok, make_float4(dwl[mm][0], dwl[mm][1], dwl[mm][2], dwl[mm][3]), f4_nan());
    }
    if (dbpart == 0) slab_db[(size_t)blockIdx.x * PD + dbc] = ok ? dbl : bad;
  };
  write_slab(dw1, db1, p.slab_dw1, p.slab_db1);
  write_slab(dw2, db2, p.slab_dw2, p.slab_db2);
}


// ---------------------------------------------------------------------------------------------------------------------
// discrete adjoint, TWO trajectories of a batch interleaved in one workgroup (see node_fwd_persistent2_kernel)
// ---------------------------------------------------------------------------------------------------------------------
// Registers are what this kernel is short of (128 per thread at two workgroups per CU; the parameter-gradient accumulators of
// both layers take 16, the matrix products ~40), so NO per-slot state stays in them between slot-phases:
//   * lambda lives in p.lam (its own array, [member][N][64]) and the stage adjoints U-bar_1..5 in a scratch array (p.ubar:
//     [slot][5][N][64]) -- rows private to the thread that owns them, written when formed, fetched at the top of the slot-phase
//     that combines them and dead again before the matrix products;
//   * a slot's own rows of the exchanged array come back by the same LDS-DMA as the foreign ones (they were stored write-through
//     a slot-phase earlier and are drained by then).
// Same products, same order of additions per member as node_bwd_persistent_kernel: du0 is bitwise equal; the parameter gradients
// are summed over the members in interleaved order (equal to rounding).
struct BNext {          // the slot-phase after this one (see fwd_slot_phase): what can be fetched for it ahead of time
  bool gather;          // it gathers halo rows (from X, after the flags of its wait list show ph - 1)
  int ph;
  const float *X;
  const unsigned *flags;
  bool tape;            // it reads a tape row and sign bits (event ev)
  size_t ev;
};

// PAIR: the slots are two TILES of one member (see fwd_slot_phase); a look at the next slot-phase's flags then never blocks -- with
// cross-slot neighbours a workgroup spinning for a flag that its neighbour publishes only behind ITS spin would be a cycle -- and
// a miss takes the blocking path at the next T0, which publishes what is pending first.
template <bool PAIR>
__global__ __launch_bounds__(kThreads, 4) void node_bwd_persistent2_kernel(const PBwdK p) {
  __shared__ __attribute__((aligned(16))) float lds[kXhF + 2 * kTileF + 2 * kWF + (PAIR ? 2 : 1) * kMetaF + 48 + 4];
  float *ldsXh = lds, *ldsG = lds, *ldsDZ = lds + kXhF, *ldsX = ldsDZ + kTileF, *ldsW1 = ldsX + kTileF, *ldsW2 = ldsW1 + kWF;
  float *ldsMeta = ldsW2 + kWF, *ldsC = ldsMeta + (PAIR ? 2 : 1) * kMetaF;
  int *s_ok = reinterpret_cast<int *>(ldsC + 48), *s_pre = s_ok + 1;
  TileCtx c, c1s;
  tile_ctx_init(p.m, c, ldsMeta, PAIR ? xcd_tile(blockIdx.x, p.pair_wgs) : -1);
  const bool has1 = !PAIR || c.tile + p.pair_wgs < p.m.n_tiles;
  if (PAIR) tile_ctx_init(p.m, c1s, ldsMeta + kMetaF, has1 ? c.tile + p.pair_wgs : c.tile);
  const TileCtx &c1 = PAIR ? c1s : c;
  if (c.tid < 48) ldsC[c.tid] = p.cb[c.tid];
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
  load_weight_lds(p.w1, ldsW1, c.tid, false);
  load_weight_lds(p.w2, ldsW2, c.tid, false);
  if (c.grp == 0) Xh4[kHaloCap * PG::LPR + c.q] = f4_zero();
  if (c.tid == 0) *s_ok = 1, *s_pre = 0;
  const unsigned own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
  const unsigned own1 = (unsigned)c1.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
  constexpr int NT = PG::CT * PG::CT;
  f32x4 dw1[PG::DWT], dw2[PG::DWT];
#pragma unroll
  for (int mm = 0; mm < PG::DWT; ++mm) dw1[mm] = dw2[mm] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float db1 = 0.f, db2 = 0.f;
  const int dbc = c.tid / PG::DBP, dbpart = c.tid % PG::DBP;
  const int S = p.S;
  __syncthreads();

  // state of the software pipeline (uniform): was the coming slot-phase's halo gathered ahead; the flag still to be published;
  // the tape row and sign bits fetched ahead for the coming slot-phase
  bool pre = false, dead = false;   // dead: a wait inside a dense half was aborted
  int n_ahead = 0;
  unsigned *pend_flags = nullptr;
  int pend_ph = 0;
  unsigned pf_mk = 0;
  float4 pf_x = f4_zero();

  // T0 of a slot-phase: everything this wave has in flight lands (halo rows gathered ahead, the previous slot-phase's row
  // stores), the workgroup meets, the previous slot-phase's flag goes out
  auto t0_publish = [&]() {
    wait_vmcnt0();
    __syncthreads();
    if (pend_flags && c.tid == 0) __hip_atomic_store(pend_flags, (unsigned)pend_ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pend_flags = nullptr;
  };

  // dWt[i][o] += sum_n A[n][i] dZ[n][o] over the tile's 32 rows; db += column sums of dZ.  The operand tiles are __restrict__ so that
  // these LDS reads do not wait for a DMA issued just before them into the halo region (halo_fill_ahead)
  auto dw_products = [&](const float *__restrict__ tX, const float *__restrict__ tDZ, f32x4 (&dwl)[PG::DWT], float &dbl) {
    const int i16 = c.lane & 15, kq = c.lane >> 4;
#pragma unroll
    for (int mm = 0; mm < PG::DWT; ++mm) {
      const int tt = c.wave_u + PG::WAVES * mm;
      if (tt < NT) {   // wave-uniform
        const int mt = tt / PG::CT, nt = tt % PG::CT;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {   // two halves of the 32-row contraction: 8 operand registers live instead of 16
          float a[kTM / 8], b[kTM / 8];
#pragma unroll
          for (int ks = 0; ks < kTM / 8; ++ks) {
            a[ks] = tX[(4 * (ks + 4 * kh) + kq) * PG::TS + mt * 16 + i16];
            b[ks] = tDZ[(4 * (ks + 4 * kh) + kq) * PG::TS + nt * 16 + i16];
          }
#pragma unroll
          for (int ks = 0; ks < kTM / 8; ++ks) dwl[mm] = mfma16(a[ks], b[ks], dwl[mm]);
        }
      }
    }
    float sdb = 0.f;
#pragma unroll
    for (int nn = dbpart; nn < kTM; nn += PG::DBP) sdb += tDZ[nn * PG::TS + dbc];
#pragma unroll
    for (int o = 1; o < PG::DBP; o <<= 1) sdb += __shfl_xor(sdb, o);
    dbl += sdb;
  };

  // the dense half of a slot-phase (T1's tail .. T5): dL/dy = c .* K-bar, relu' by the sign bits, G = dZ W^T -> c .* G stored for the
  // next gather (not drained: the flag goes out at the next T0), dW += A^T dZ, db += column sums; the next slot-phase's flags are
  // looked at under the matrix products and, if they are all there, its halo rows go out before the row stores
  // (c, own: this slot's tile; cn, ownn: the next slot-phase's)
  auto dense = [&](const TileCtx &c, const TileCtx &cn, unsigned own, unsigned ownn, unsigned *flags, int ph, const float *ldsW,
                   f32x4 (&dwl)[PG::DWT], float &dbl, float4 kbar, unsigned mk, float4 xrow, float *gout, const BNext &nx) {
    kbar = f4_scale(c.ci, kbar);
    const float4 dz = c.valid ? make_float4((mk & 1u) ? kbar.x : 0.f, (mk & 2u) ? kbar.y : 0.f, (mk & 4u) ? kbar.z : 0.f,
                                            (mk & 8u) ? kbar.w : 0.f)
                              : f4_zero();
    *reinterpret_cast<float4 *>(&ldsDZ[c.grp * PG::TS + 4 * c.q]) = dz;
    *reinterpret_cast<float4 *>(&ldsX[c.grp * PG::TS + 4 * c.q]) = f4_sel(c.valid, xrow, f4_zero());
    __syncthreads();   // T2
    if (nx.tape) {     // the next slot-phase's tape row and sign bits: a slot-phase ahead, so that they are there at its T0
      pf_mk = ldu8_g(p.masks + nx.ev * p.mask_bytes + (size_t)cn.tile * kThreads, (unsigned)c.tid);
      pf_x = ld4_stream_g(p.tape + nx.ev * p.row_elems, ownn);
    }
    NGPDE_PST(p.m, ph, 3);
    // T3: both matrix products of the phase back to back (no barrier between them: they read the same operand tiles and write
    // different things): G = dZ W^T into the halo region (free since T2), then dW += A^T dZ, db.  Wave 0 fetches the next
    // slot-phase's flags between the two -- as late as it can be done with the answer still there when the products end
    mfma_rows_times_bt<PD>(ldsDZ, ldsW, ldsG, c.wave_u, c.lane);
    unsigned f1 = 0;
    if (nx.gather && c.wave_u == 0) f1 = poll_issue(p.m, cn, nx.flags);
    dw_products(ldsX, ldsDZ, dwl, dbl);
    // T4: the next slot-phase's flags (members: wave 0 spins if this workgroup leads its neighbours, see tile_wait_primed; tile
    // pairs: one look, see the kernel's head), the barrier
    if (nx.gather && !PAIR) {   // uniform
      if (!tile_wait_primed(p.m, cn, nx.ph, s_ok, nx.flags, f1)) { dead = true; return; }
      pre = true;
    } else if (nx.gather) {
      if (c.wave_u == 0) {
        const bool hit = poll_ready(cn, f1, nx.ph);
        if (c.lane == 0) *s_pre = hit ? 1 : 0;
      }
      __syncthreads();
      pre = *s_pre != 0;
    } else {
      __syncthreads();
      pre = false;
    }
    NGPDE_PST(p.m, ph, 4);
    const float4 gv = f4_sel(c.valid, f4_scale(c.ci, *reinterpret_cast<const float4 *>(&ldsG[c.grp * PG::TS + 4 * c.q])), f4_zero());
    if (pre) {   // uniform
      __syncthreads();   // every thread has read its row of G: the region is the halo again
      halo_fill_all(cn, nx.X, ldsXh);
    }
    if (c.valid) store_sc1(gout, own, gv);
    NGPDE_PST(p.m, ph, 5);
    pend_flags = flags + 32 * c.tile;
    pend_ph = ph;
  };

  // the gathering half: T0, then (unless the rows were gathered ahead) own rows + blocking wait + gather
  auto top = [&](const TileCtx &c, const unsigned *flags, int ph, const float *X) -> bool {
    t0_publish();
    n_ahead += pre ? 1 : 0;
    if (!pre) {
      if (!tile_wait(p.m, c, ph, s_ok, flags)) return false;
      NGPDE_PST(p.m, ph, 1);
      halo_fill_all(c, X, ldsXh);
      wait_vmcnt0();
      __syncthreads();
    }
    NGPDE_PST(p.m, ph, 2);
    return true;
  };

  // layer 1 of stage i of one slot: dL/dy1 = A^T g2
  auto phase_l1 = [&](const TileCtx &c, const TileCtx &cn, unsigned own, unsigned ownn, int sl, int ph, const BNext &nx) -> bool {
    float *g1 = p.g1 + (PAIR ? 0 : (size_t)sl * p.row_elems), *g2 = p.g2 + (PAIR ? 0 : (size_t)sl * p.row_elems);
    unsigned *flags = p.m.flags + (PAIR ? 0 : (size_t)sl * p.flag_stride);
    NGPDE_PST(p.m, ph, 0);
    const unsigned mk = pf_mk;
    const float4 xrow = pf_x;
    if (!top(c, flags, ph, g2)) return false;
    const float4 t = tile_aggregate_lean(c, ldsXh);   // (the unrolled form with the slot words in registers: 24 spilled dwords)
    dense(c, cn, own, ownn, flags, ph, ldsW1, dw1, db1, t, mk, xrow, g1, nx);
    return !dead;
  };

  // U-bar_i = A^T g1; K-bar of the stage evaluated before it (or the lambda update), layer 2's dense half.  `last`: the member's
  // final phase (no dense half, nothing published)
  auto phase_l2 = [&](const TileCtx &c, const TileCtx &cn, unsigned own, unsigned ownn, int sl, float *lam_g, int ph, int i, bool last,
                      const BNext &nx) -> bool {
    float *g1 = p.g1 + (PAIR ? 0 : (size_t)sl * p.row_elems), *g2 = p.g2 + (PAIR ? 0 : (size_t)sl * p.row_elems);
    float *ubar = p.ubar + (PAIR ? 0 : (size_t)sl * 5 * p.row_elems);   // (tile pairs: the rows of one member)
    unsigned *flags = p.m.flags + (PAIR ? 0 : (size_t)sl * p.flag_stride);
    NGPDE_PST(p.m, ph, 0);
    const unsigned mk = pf_mk;
    const float4 xrow = pf_x;
    if (!top(c, flags, ph, g1)) return false;
    // the stage adjoints this combination needs: U-bar_j, j > i, of THIS step (j < S); everything else enters as an exact zero
    // (node_bwd_persistent_kernel keeps zeros / finished values with zero weights in those places).  ONE scalar base + 32-bit
    // offsets the optimiser cannot see through: five bases per slot ended up as 64-bit vector addresses hoisted out of the loops
    // and spilled
    float4 ub1 = f4_zero(), ub2 = f4_zero(), ub3 = f4_zero(), ub4 = f4_zero(), ub5 = f4_zero();
    const unsigned rowb = (unsigned)(p.row_elems * sizeof(float));
    unsigned uo = own;
    asm volatile("" : "+v"(uo));
    const float4 lam = f4_sel(c.valid, ld4_g(lam_g, uo), f4_zero());
    if (i < 1 && 1 < S) ub1 = ld4_g(ubar, uo);
    if (i < 2 && 2 < S) ub2 = ld4_g(ubar, uo + rowb);
    if (i < 3 && 3 < S) ub3 = ld4_g(ubar, uo + 2 * rowb);
    if (i < 4 && 4 < S) ub4 = ld4_g(ubar, uo + 3 * rowb);
    if (i < 5 && 5 < S) ub5 = ld4_g(ubar, uo + 4 * rowb);
    const float4 t = tile_aggregate_lean(c, ldsXh);   // (the unrolled form with the slot words in registers: 24 spilled dwords)
    float4 kbar;
    if (i >= 1) {
      if (c.valid) st4_g(ubar, uo + (unsigned)(i - 1) * rowb, t);   // U-bar_i
      float4 v = f4_scale(ldsC[42 + i], t);
      v = f4_fma(ldsC[i - 1], lam, v);
      v = f4_fma(ldsC[6 + i * 6 + 2], ub2, v); v = f4_fma(ldsC[6 + i * 6 + 3], ub3, v);
      v = f4_fma(ldsC[6 + i * 6 + 4], ub4, v); v = f4_fma(ldsC[6 + i * 6 + 5], ub5, v);
      kbar = v;
    } else {
      float4 v = f4_scale(1.0f, t);
      v = f4_fma(1.0f, lam, v);
      v = f4_fma(1.0f, ub1, v); v = f4_fma(1.0f, ub2, v); v = f4_fma(1.0f, ub3, v);
      v = f4_fma(1.0f, ub4, v); v = f4_fma(1.0f, ub5, v);
      if (c.valid) st4_g(lam_g, uo, v);   // lambda of the step before (the member's dL/du~0 at the end)
      kbar = f4_scale(ldsC[S - 1], v);
    }
    if (!last) dense(c, cn, own, ownn, flags, ph, ldsW2, dw2, db2, kbar, mk, xrow, g2, nx);
    else pre = false;
    return !dead;
  };

  bool ok = true;
  int ph = 0;
  const size_t per = (size_t)p.n_steps * S * 2;   // tape events of one member
  for (int mb = 0; mb < p.n_members && ok; mb += 2) {
    const bool two = PAIR ? has1 : mb + 1 < p.n_members;
    float *lam_g0 = p.lam + (size_t)mb * p.row_elems, *lam_g1 = lam_g0 + ((two && !PAIR) ? p.row_elems : 0);
    const size_t ev00 = (size_t)mb * per, ev01 = PAIR ? ev00 : ev00 + per;
    const size_t goff1 = PAIR ? 0 : p.row_elems;            // slot 1's offset in the exchanged arrays
    unsigned *flags0 = p.m.flags, *flags1 = p.m.flags + (PAIR ? 0 : p.flag_stride);
    const size_t e1_first = (size_t)((p.n_steps - 1) * S + (S - 1)) * 2;   // layer-1 event of the first stage the adjoint visits
    {   // first phase of a member: K-bar of the last stage of the last step = dt b_S lambda, layer 2's dense half (no gather, no wait)
      ++ph;
      const size_t e = e1_first + 1;
      BNext nx;
      {
        t0_publish();
        const unsigned mk = ldu8_g(p.masks + (ev00 + e) * p.mask_bytes + (size_t)c.tile * kThreads, (unsigned)c.tid);
        const float4 xrow = ld4_stream_g(p.tape + (ev00 + e) * p.row_elems, own);
        // next: the same phase of slot 1 (nothing to gather; its tape row is read there), or layer 1 of slot 0
        // (one slot: nothing is gathered ahead -- the next slot-phase is this slot's own next phase, whose flags cannot be there
        // before this one's is published -- only its tape row is fetched)
        nx.gather = false; nx.ph = ph + 1; nx.X = p.g2; nx.flags = flags0; nx.tape = !two; nx.ev = ev00 + e1_first;
        const float4 lam = f4_sel(c.valid, ld4_g(lam_g0, own), f4_zero());
        dense(c, two ? c1 : c, own, two ? own1 : own, flags0, ph, ldsW2, dw2, db2, f4_scale(ldsC[S - 1], lam), mk, xrow, p.g2, nx);
        if (dead) ok = false;
      }
      if (two && ok) {
        t0_publish();
        const unsigned mk = ldu8_g(p.masks + (ev01 + e) * p.mask_bytes + (size_t)c1.tile * kThreads, (unsigned)c.tid);
        const float4 xrow = ld4_stream_g(p.tape + (ev01 + e) * p.row_elems, own1);
        nx.gather = true; nx.ph = ph + 1; nx.X = p.g2; nx.flags = flags0; nx.tape = true; nx.ev = ev00 + e1_first;
        const float4 lam = f4_sel(c1.valid, ld4_g(lam_g1, own1), f4_zero());
        dense(c1, c, own1, own, flags1, ph, ldsW2, dw2, db2, f4_scale(ldsC[S - 1], lam), mk, xrow, p.g2 + goff1, nx);
        if (dead) ok = false;
      }
    }
    for (int n = p.n_steps - 1; n >= 0 && ok; --n) {
      for (int i = S - 1; i >= 0 && ok; --i) {
        const bool last = (i == 0 && n == 0);
        const size_t e1 = (size_t)(n * S + i) * 2;
        const size_t e2 = (i >= 1) ? (size_t)(n * S + i - 1) * 2 + 1 : (size_t)((max(n, 1) - 1) * S + (S - 1)) * 2 + 1;
        const size_t e1n = (i >= 1) ? (size_t)(n * S + i - 1) * 2 : (size_t)((max(n, 1) - 1) * S + (S - 1)) * 2;   // layer-1 event of the stage after this one
        BNext nx;
        ++ph;   // layer 1
        // after (L1, slot 0): (L1, slot 1) or, with one slot, (L2, slot 0); after (L1, slot 1): (L2, slot 0)
        nx.gather = two; nx.ph = two ? ph : ph + 1; nx.X = two ? p.g2 + goff1 : p.g1; nx.flags = two ? flags1 : flags0;
        nx.tape = two ? true : !last; nx.ev = two ? ev01 + e1 : ev00 + e2;
        if (!phase_l1(c, two ? c1 : c, own, two ? own1 : own, 0, ph, nx)) { ok = false; break; }
        if (two) {
          nx.gather = true; nx.ph = ph + 1; nx.X = p.g1; nx.flags = flags0; nx.tape = !last; nx.ev = ev00 + e2;
          if (!phase_l1(c1, c, own1, own, 1, ph, nx)) { ok = false; break; }
        }
        ++ph;   // layer 2
        // after (L2, slot 0): (L2, slot 1) or, with one slot, the next stage's (L1, slot 0); after (L2, slot 1): the next (L1, slot 0)
        nx.gather = two; nx.ph = two ? ph : ph + 1; nx.X = two ? p.g1 + goff1 : p.g2; nx.flags = two ? flags1 : flags0;
        nx.tape = two ? !last : !last; nx.ev = two ? ev01 + e2 : ev00 + e1n;
        if (!phase_l2(c, two ? c1 : c, own, two ? own1 : own, 0, lam_g0, ph, i, last, nx)) { ok = false; break; }
        if (two) {
          nx.gather = !last; nx.ph = ph + 1; nx.X = p.g2; nx.flags = flags0; nx.tape = !last; nx.ev = ev00 + e1n;
          if (!phase_l2(c1, c, own1, own, 1, lam_g1, ph, i, last, nx)) { ok = false; break; }
        }
      }
    }
    // the last published slot-phase's flag (the final phase of a member publishes nothing; the next pair's first phase publishes
    // ph + 1 and waits for nothing)
    t0_publish();
    if (PAIR) break;   // one member
  }
  if (!ok) {
    for (int mb = 0; mb < p.n_members; ++mb) {
      if (c.valid) st4_g(p.lam + (size_t)mb * p.row_elems, own, f4_nan());
      if (PAIR && has1 && c1.valid) st4_g(p.lam + (size_t)mb * p.row_elems, own1, f4_nan());
    }
  }
  const float bad = __int_as_float(0x7fc00000);
  auto write_slab = [&](const f32x4 (&dwl)[PG::DWT], float dbl, float *slab_dw, float *slab_db) {
    float4 *slab4 = reinterpret_cast<float4 *>(slab_dw + (size_t)blockIdx.x * PD * PD);
#pragma unroll
    for (int mm = 0; mm < PG::DWT; ++mm) {
      const int tt = c.wave_u + PG::WAVES * mm;
      if (tt < NT) slab4[tt * 64 + c.lane] = f4_sel(ok, make_float4(dwl[mm][0], dwl[mm][1], dwl[mm][2], dwl[mm][3]), f4_nan());
    }
    if (dbpart == 0) slab_db[(size_t)blockIdx.x * PD + dbc] = ok ? dbl : bad;
  };
  write_slab(dw1, db1, p.slab_dw1, p.slab_db1);
  write_slab(dw2, db2, p.slab_dw2, p.slab_db2);
  if (c.tid == 0 && p.m.stats) p.m.stats[2 * c.tile + 1] = n_ahead;
}

// ---------------------------------------------------------------------------------------------------------------------
// adjoint, K tiles per workgroup taking turns (see node_fwd_persistentK_kernel): lambda and the stage adjoints are own rows of
// p.lam / p.ubar (zero at launch where the one-tile kernel starts from zero registers), one layer's W in LDS at a time, the
// parameter-gradient accumulators in registers over all tiles and phases (one slab per WORKGROUP at the end).
// ---------------------------------------------------------------------------------------------------------------------
template <int ACT, bool WGT = false>
__global__ __launch_bounds__(kThreads, 4) void node_bwd_persistentK_kernel(const PBwdK p) {
  constexpr bool RELU = (ACT == NGPDE_ACT_RELU);
  using Aux = typename std::conditional<RELU, unsigned, float4>::type;
  constexpr int kMS = meta_stride<WGT>(), kMT = meta_tiles<WGT>();
  __shared__ __attribute__((aligned(16))) float lds[kXhF + 2 * kTileF + kWF + kMT * kMS + 48 + 4];
  static_assert(sizeof(lds) <= 80 * 1024 - 64, "two workgroups per CU");
  float *ldsXh = lds, *ldsG = lds, *ldsDZ = lds + kXhF, *ldsX = ldsDZ + kTileF, *ldsW = ldsX + kTileF;
  float *ldsMeta = ldsW + kWF, *ldsC = ldsMeta + kMT * kMS;
  int *s_ok = reinterpret_cast<int *>(ldsC + 48);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 48) ldsC[tid] = p.cb[tid];
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
  if (tid < PG::LPR) Xh4[kHaloCap * PG::LPR + tid] = f4_zero();
  if (tid == 0) *s_ok = 1;
  constexpr int NT = PG::CT * PG::CT;
  f32x4 dw1[PG::DWT], dw2[PG::DWT];
#pragma unroll
  for (int mm = 0; mm < PG::DWT; ++mm) dw1[mm] = dw2[mm] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float db1 = 0.f, db2 = 0.f;
  const int dbc = tid / PG::DBP, dbpart = tid % PG::DBP;
  const int S = p.S, W = p.pair_wgs, K = min(p.k_tiles, kMT);
  const int t0 = xcd_tile(blockIdx.x, W);
  const unsigned rowb = (unsigned)(p.row_elems * sizeof(float));
  for (int s = 0; s < K && t0 + s * W < p.m.n_tiles; ++s) tile_tables_to_lds<WGT>(p.m, t0 + s * W, ldsMeta + s * kMS);
  __syncthreads();

  // the dense half of a turn (node_bwd_persistent_kernel's, with the tile context as an argument)
  // The tape row and sign bits of the NEXT turn (mk_n / xrow_n: the workgroup's next tile in this phase, or its first tile in the next
  // phase) are asked for at the head of this turn's dense half, when this turn's own have just gone to LDS: cold rows, ~4.5 k cycles
  // away, with the product, the stores, the drain and the parameter-gradient products to arrive in.  (The drain in front of the flag
  // waits for them, which costs the publish ~1.5 k cycles -- nobody reads a tile's rows before the workgroup's other turns are through.
  // Asked for at the head of their own turn, in front of the poll, they held every turn's first look at the flags up by their
  // latency: a wave's loads return in order.)
  Aux mk_n{};
  float4 xrow_n = f4_zero();
  int Kv = 0;
  for (int s = 0; s < K && t0 + s * W < p.m.n_tiles; ++s) Kv = s + 1;
  auto fetch = [&](size_t ev, int s) {
    const int node = max(reinterpret_cast<const int4 *>(ldsMeta + s * kMS + kMetaF)[tid >> 4].x, 0);
    const unsigned own = (unsigned)node * (unsigned)(PD * 4) + (unsigned)((tid & 15) * 16);
    if constexpr (RELU) mk_n = ldu8_g(p.masks + ev * p.mask_bytes + (size_t)(t0 + s * W) * kThreads, (unsigned)tid);
    else mk_n = ld4_stream_g(p.ztape + ev * p.row_elems, own);
    xrow_n = ld4_stream_g(p.tape + ev * p.row_elems, own);
  };
  auto dense = [&](const TileCtx &c, unsigned own, int ph, f32x4 (&dwl)[PG::DWT], float &dbl, float4 kbar, Aux mk, float4 xrow, float *gout,
                   bool pf, size_t ev_n, int s_n) {
    kbar = f4_scale(c.ci, kbar);
    float4 dz;
    if constexpr (RELU) {
      dz = c.valid ? make_float4((mk & 1u) ? kbar.x : 0.f, (mk & 2u) ? kbar.y : 0.f, (mk & 4u) ? kbar.z : 0.f, (mk & 8u) ? kbar.w : 0.f)
                   : f4_zero();
    } else {
      dz = f4_sel(c.valid, f4_mul(kbar, f4_dact(p.act, mk)), f4_zero());
    }
    *reinterpret_cast<float4 *>(&ldsDZ[c.grp * PG::TS + 4 * c.q]) = dz;
    *reinterpret_cast<float4 *>(&ldsX[c.grp * PG::TS + 4 * c.q]) = f4_sel(c.valid, xrow, f4_zero());
    if (pf) fetch(ev_n, s_n);
    __syncthreads();
    mfma_rows_times_bt<PD>(ldsDZ, ldsW, ldsG, c.wave_u, c.lane);
    __syncthreads();
    const float4 gv = f4_sel(c.valid, f4_scale(c.ci, *reinterpret_cast<const float4 *>(&ldsG[c.grp * PG::TS + 4 * c.q])), f4_zero());
    if (c.valid) store_sc1(gout, own, gv);
    tile_publish(p.m, c, ph);
    const int i16 = c.lane & 15, kq = c.lane >> 4;
#pragma unroll
    for (int mm = 0; mm < PG::DWT; ++mm) {
      const int tt = c.wave_u + PG::WAVES * mm;
      if (tt < NT) {   // wave-uniform
        const int mt = tt / PG::CT, nt = tt % PG::CT;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          float a[kTM / 8], b[kTM / 8];
#pragma unroll
          for (int ks = 0; ks < kTM / 8; ++ks) {
            a[ks] = ldsX[(4 * (ks + 4 * kh) + kq) * PG::TS + mt * 16 + i16];
            b[ks] = ldsDZ[(4 * (ks + 4 * kh) + kq) * PG::TS + nt * 16 + i16];
          }
#pragma unroll
          for (int ks = 0; ks < kTM / 8; ++ks) dwl[mm] = mfma16(a[ks], b[ks], dwl[mm]);
        }
      }
    }
    {
      float s = 0.f;
#pragma unroll
      for (int nn = dbpart; nn < kTM; nn += PG::DBP) s += ldsDZ[nn * PG::TS + dbc];
#pragma unroll
      for (int o = 1; o < PG::DBP; o <<= 1) s += __shfl_xor(s, o);
      dbl += s;
    }
  };

  bool ok = true;
  int ph = 0;
  {   // first phase: K-bar of the last stage of the last step = dt b_S lambda, layer 2's dense half (no gather, no wait)
    ++ph;
    const size_t ev = (size_t)((p.n_steps - 1) * S + (S - 1)) * 2 + 1;
    load_weight_lds(p.w2, ldsW, tid, false);
    if (Kv > 0) fetch(ev, 0);
    for (int s = 0; s < K; ++s) {
      const int tile = t0 + s * W;
      if (tile >= p.m.n_tiles) break;
      TileCtx c;
      tile_ctx_from_lds<WGT>(c, tile, ldsMeta + s * kMS);
      const unsigned own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
      __syncthreads();   // (the phase's W is in LDS; the previous turn's products are done with the operand tiles)
      const float4 lam = f4_sel(c.valid, ld4_g(p.lam, own), f4_zero());
      const Aux mk = mk_n;
      const float4 xrow = xrow_n;
      const bool more = s + 1 < Kv;   // next: the workgroup's next tile, or layer 1 of the last stage of the last step on its first one
      dense(c, own, ph, dw2, db2, f4_scale(ldsC[S - 1], lam), mk, xrow, p.g2, true,
            more ? ev : (size_t)((p.n_steps - 1) * S + (S - 1)) * 2, more ? s + 1 : 0);
    }
  }
  for (int n = p.n_steps - 1; n >= 0 && ok; --n) {
    for (int i = S - 1; i >= 0 && ok; --i) {
      {   // layer 1 of stage i: dL/dy1 = A^T g2
        ++ph;
        const size_t ev = (size_t)(n * S + i) * 2;
        __syncthreads();
        load_weight_lds(p.w1, ldsW, tid, false);
        for (int s = 0; s < K; ++s) {
          const int tile = t0 + s * W;
          if (tile >= p.m.n_tiles) break;
          TileCtx c;
          tile_ctx_from_lds<WGT>(c, tile, ldsMeta + s * kMS);
          const unsigned own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
          const Aux mk = mk_n;
          const float4 xrow = xrow_n;
          if (!tile_wait(p.m, c, ph, s_ok)) { ok = false; break; }
          halo_fill_all(c, p.g2, ldsXh);
          wait_vmcnt0();
          __syncthreads();
          const float4 t = tile_aggregate_rounds<WGT>(c, ldsXh);
          __syncthreads();   // every thread has its sum: the region becomes the product's result tile
          // next: the workgroup's next tile, or layer 2 of the stage evaluated before this one (the very last phase asks for nothing)
          const bool more = s + 1 < Kv;
          const size_t ev_b = (i >= 1) ? (size_t)(n * S + i - 1) * 2 + 1 : (size_t)((max(n, 1) - 1) * S + (S - 1)) * 2 + 1;
          dense(c, own, ph, dw1, db1, t, mk, xrow, p.g1, more || !(i == 0 && n == 0), more ? ev : ev_b, more ? s + 1 : 0);
        }
        if (!ok) break;
      }
      {   // U-bar_i = A^T g1; K-bar of the stage evaluated before it (or the lambda update), layer 2's dense half
        ++ph;
        const bool last = (i == 0 && n == 0);
        const size_t ev = (i >= 1) ? (size_t)(n * S + i - 1) * 2 + 1 : (size_t)((max(n, 1) - 1) * S + (S - 1)) * 2 + 1;
        __syncthreads();
        load_weight_lds(p.w2, ldsW, tid, false);
        for (int s = 0; s < K; ++s) {
          const int tile = t0 + s * W;
          if (tile >= p.m.n_tiles) break;
          TileCtx c;
          tile_ctx_from_lds<WGT>(c, tile, ldsMeta + s * kMS);
          const unsigned own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
          const Aux mk = mk_n;        // (the last phase runs no dense half: nothing was asked for)
          const float4 xrow = xrow_n;
          if (!tile_wait(p.m, c, ph, s_ok)) { ok = false; break; }
          halo_fill_all(c, p.g1, ldsXh);
          // lambda and the stage adjoints of this step from memory (rows 0..4 of ubar = U-bar_1..5), in flight under the gather
          const float4 lam = f4_sel(c.valid, ld4_g(p.lam, own), f4_zero());
          // (only the stage adjoints this stage combines, U-bar_j with i < j < S: the others meet a zero coefficient or are zero)
          const float4 lb1 = (i < 1 && S > 1) ? ld4_g(p.ubar, own) : f4_zero(), lb2 = (i < 2 && S > 2) ? ld4_g(p.ubar, own + rowb) : f4_zero(),
                       lb3 = (i < 3 && S > 3) ? ld4_g(p.ubar, own + 2 * rowb) : f4_zero(), lb4 = (i < 4 && S > 4) ? ld4_g(p.ubar, own + 3 * rowb) : f4_zero(),
                       lb5 = (i < 5 && S > 5) ? ld4_g(p.ubar, own + 4 * rowb) : f4_zero();
          wait_vmcnt0();
          __syncthreads();
          const float4 t = tile_aggregate_rounds<WGT>(c, ldsXh);
          __syncthreads();
          const float4 ub1 = i == 1 ? t : lb1, ub2 = i == 2 ? t : lb2, ub3 = i == 3 ? t : lb3, ub4 = i == 4 ? t : lb4,
                       ub5 = i == 5 ? t : lb5;   // U-bar_i is t itself
          float4 kbar;
          if (i >= 1) {
            if (c.valid) st4_g(p.ubar, own + (unsigned)(i - 1) * rowb, t);
            float4 v = f4_scale(ldsC[42 + i], t);
            v = f4_fma(ldsC[i - 1], lam, v);
            v = f4_fma(ldsC[6 + i * 6 + 2], ub2, v); v = f4_fma(ldsC[6 + i * 6 + 3], ub3, v);
            v = f4_fma(ldsC[6 + i * 6 + 4], ub4, v); v = f4_fma(ldsC[6 + i * 6 + 5], ub5, v);
            kbar = v;
          } else {
            float4 v = f4_scale(1.0f, t);
            v = f4_fma(1.0f, lam, v);
            v = f4_fma(1.0f, ub1, v); v = f4_fma(1.0f, ub2, v); v = f4_fma(1.0f, ub3, v);
            v = f4_fma(1.0f, ub4, v); v = f4_fma(1.0f, ub5, v);
            if (c.valid) st4_g(p.lam, own, v);
            kbar = f4_scale(ldsC[S - 1], v);
          }
          if (!last) {   // next: the workgroup's next tile, or layer 1 of stage i - 1 / of the last stage of the step before
            const bool more = s + 1 < Kv;
            const size_t ev_a = (size_t)(i >= 1 ? n * S + i - 1 : (max(n, 1) - 1) * S + (S - 1)) * 2;
            dense(c, own, ph, dw2, db2, kbar, mk, xrow, p.g2, true, more ? ev : ev_a, more ? s + 1 : 0);
          } else {
            __syncthreads();
          }
        }
        if (!ok) break;
      }
    }
  }
  if (!ok) {
    __syncthreads();
    for (int s = 0; s < K; ++s) {
      const int tile = t0 + s * W;
      if (tile >= p.m.n_tiles) break;
      const int4 sc = p.m.sched[(size_t)tile * kTM + (tid >> 4)];
      if (sc.x >= 0) st4_g(p.lam, (unsigned)sc.x * (unsigned)(PD * 4) + (unsigned)((tid & 15) * 16), f4_nan());
    }
  }
  const float bad = __int_as_float(0x7fc00000);
  auto write_slab = [&](const f32x4 (&dwl)[PG::DWT], float dbl, float *slab_dw, float *slab_db) {
    float4 *slab4 = reinterpret_cast<float4 *>(slab_dw + (size_t)blockIdx.x * PD * PD);
#pragma unroll
    for (int mm = 0; mm < PG::DWT; ++mm) {
      const int tt = wave_u + PG::WAVES * mm;
      if (tt < NT) slab4[tt * 64 + lane] = f4_sel(ok, make_float4(dwl[mm][0], dwl[mm][1], dwl[mm][2], dwl[mm][3]), f4_nan());
    }
    if (dbpart == 0) slab_db[(size_t)blockIdx.x * PD + dbc] = ok ? dbl : bad;
  };
  write_slab(dw1, db1, p.slab_dw1, p.slab_db1);
  write_slab(dw2, db2, p.slab_dw2, p.slab_db2);
}


// ---------------------------------------------------------------------------------------------------------------------
// adjoint, tile rounds with the NEXT turn's hand-off taken off the current turn's instruction stream (round 5)
// ---------------------------------------------------------------------------------------------------------------------
// node_bwd_persistentK_kernel runs poll -> gather -> aggregate -> product -> store -> drain -> flag -> parameter-gradient products one
// after the other for every turn.  But a workgroup's NEXT turn (its next tile in this phase, or its first tile in the next phase)
// depends on nothing this turn produces, so -- the tile-pair kernel's pipeline (node_bwd_persistent2_kernel<true>), for K tiles:
//   T0  s_waitcnt vmcnt(0) + barrier: this turn's halo rows (gathered during the previous turn's parameter-gradient products) have
//       landed, the previous turn's row stores are drained -> ITS flag goes out here (deferred publish)
//   T1  LDS aggregation, K-bar, operand tiles; the next turn's tape row and sign bits are asked for; barrier; G = dZ W^T
//   T2  wave 0 asks for the flags of the next turn's wait list, not waited for; dW += A^T dZ, db (the longest stretch of the turn)
//   T3  one look at the flags; if they were all there, the next turn's rows go out by LDS-DMA behind this turn's read of G (the halo
//       region is the product's result tile); row stores (not drained)
// A next turn whose flags were not there (the first tile of the next phase, mostly: its neighbours are in THIS phase) takes the
// blocking path at its T0, which publishes what is pending first -- so a workgroup never spins in front of its own publish.
// Arithmetic per tile is the tile-round kernel's, operation for operation: du0 bitwise equal, NGPDE_NO_TILE_PIPE=1 selects it.
template <int ACT>
__global__ __launch_bounds__(kThreads, 4) void node_bwd_persistentKP_kernel(const PBwdK p) {
  constexpr bool RELU = (ACT == NGPDE_ACT_RELU);
  using Aux = typename std::conditional<RELU, unsigned, float4>::type;
  constexpr int kMS = meta_stride<false>(), kMT = meta_tiles<false>();
  __shared__ __attribute__((aligned(16))) float lds[kXhF + 2 * kTileF + kWF + kMT * kMS + 48 + 4];
  static_assert(sizeof(lds) <= 80 * 1024 - 64, "two workgroups per CU");
  float *ldsXh = lds, *ldsG = lds, *ldsDZ = lds + kXhF, *ldsX = ldsDZ + kTileF, *ldsW = ldsX + kTileF;
  float *ldsMeta = ldsW + kWF, *ldsC = ldsMeta + kMT * kMS;
  int *s_ok = reinterpret_cast<int *>(ldsC + 48), *s_pre = s_ok + 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 48) ldsC[tid] = p.cb[tid];
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
  if (tid < PG::LPR) Xh4[kHaloCap * PG::LPR + tid] = f4_zero();
  if (tid == 0) *s_ok = 1, *s_pre = 0;
  constexpr int NT = PG::CT * PG::CT;
  f32x4 dw1[PG::DWT], dw2[PG::DWT];
#pragma unroll
  for (int mm = 0; mm < PG::DWT; ++mm) dw1[mm] = dw2[mm] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float db1 = 0.f, db2 = 0.f;
  const int dbc = tid / PG::DBP, dbpart = tid % PG::DBP;
  const int S = p.S, W = p.pair_wgs, K = min(p.k_tiles, kMT);
  const int t0 = xcd_tile(blockIdx.x, W);
  const unsigned rowb = (unsigned)(p.row_elems * sizeof(float));
  int Kv = 0;   // tiles this workgroup really holds
  for (int s = 0; s < K && t0 + s * W < p.m.n_tiles; ++s, ++Kv) tile_tables_to_lds<false>(p.m, t0 + s * W, ldsMeta + s * kMS);
  __syncthreads();

  bool pre = false, ok = true;
  unsigned *pend_flags = nullptr;
  int pend_ph = 0, n_ahead = 0;
  Aux mk_n{};
  float4 xrow_n = f4_zero();
  struct Next {          // the turn after this one
    bool gather;         // it gathers halo rows from X once the flags of its wait list show ph - 1
    int ph, s;
    const float *X;
    bool tape;           // it runs a dense half: tape row and sign bits of event ev
    size_t ev;
  };
  auto fetch = [&](size_t ev, int s) {
    const int node = max(reinterpret_cast<const int4 *>(ldsMeta + s * kMS + kMetaF)[tid >> 4].x, 0);
    const unsigned own = (unsigned)node * (unsigned)(PD * 4) + (unsigned)((tid & 15) * 16);
    if constexpr (RELU) mk_n = ldu8_g(p.masks + ev * p.mask_bytes + (size_t)(t0 + s * W) * kThreads, (unsigned)tid);
    else mk_n = ld4_stream_g(p.ztape + ev * p.row_elems, own);
    xrow_n = ld4_stream_g(p.tape + ev * p.row_elems, own);
  };
  auto t0_publish = [&]() {
    wait_vmcnt0();
    __syncthreads();
    if (pend_flags && tid == 0) __hip_atomic_store(pend_flags, (unsigned)pend_ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pend_flags = nullptr;
  };
  // T0, then (unless the rows were gathered ahead) blocking wait + gather
  auto top = [&](const TileCtx &c, int ph, const float *X) -> bool {
    t0_publish();
    n_ahead += pre ? 1 : 0;
    if (!pre) {
      if (!tile_wait(p.m, c, ph, s_ok)) return false;
      halo_fill_all(c, X, ldsXh);
      wait_vmcnt0();
      __syncthreads();
    }
    return true;
  };
  // (operand tiles __restrict__: these LDS reads must not wait for a DMA issued into the halo region)
  auto dw_products = [&](const float *__restrict__ tX, const float *__restrict__ tDZ, f32x4 (&dwl)[PG::DWT], float &dbl) {
    const int i16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int mm = 0; mm < PG::DWT; ++mm) {
      const int tt = wave_u + PG::WAVES * mm;
      if (tt < NT) {   // wave-uniform
        const int mt = tt / PG::CT, nt = tt % PG::CT;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          float a[kTM / 8], b[kTM / 8];
#pragma unroll
          for (int ks = 0; ks < kTM / 8; ++ks) {
            a[ks] = tX[(4 * (ks + 4 * kh) + kq) * PG::TS + mt * 16 + i16];
            b[ks] = tDZ[(4 * (ks + 4 * kh) + kq) * PG::TS + nt * 16 + i16];
          }
#pragma unroll
          for (int ks = 0; ks < kTM / 8; ++ks) dwl[mm] = mfma16(a[ks], b[ks], dwl[mm]);
        }
      }
    }
    float sdb = 0.f;
#pragma unroll
    for (int nn = dbpart; nn < kTM; nn += PG::DBP) sdb += tDZ[nn * PG::TS + dbc];
#pragma unroll
    for (int o = 1; o < PG::DBP; o <<= 1) sdb += __shfl_xor(sdb, o);
    dbl += sdb;
  };
  auto dense = [&](const TileCtx &c, unsigned own, int ph, f32x4 (&dwl)[PG::DWT], float &dbl, float4 kbar, Aux mk, float4 xrow, float *gout,
                   const Next &nx) {
    kbar = f4_scale(c.ci, kbar);
    float4 dz;
    if constexpr (RELU) {
      dz = c.valid ? make_float4((mk & 1u) ? kbar.x : 0.f, (mk & 2u) ? kbar.y : 0.f, (mk & 4u) ? kbar.z : 0.f, (mk & 8u) ? kbar.w : 0.f)
                   : f4_zero();
    } else {
      dz = f4_sel(c.valid, f4_mul(kbar, f4_dact(p.act, mk)), f4_zero());
    }
    *reinterpret_cast<float4 *>(&ldsDZ[c.grp * PG::TS + 4 * c.q]) = dz;
    *reinterpret_cast<float4 *>(&ldsX[c.grp * PG::TS + 4 * c.q]) = f4_sel(c.valid, xrow, f4_zero());
    if (nx.tape) fetch(nx.ev, nx.s);
    __syncthreads();
    mfma_rows_times_bt<PD>(ldsDZ, ldsW, ldsG, wave_u, lane);
    TileCtx cn;
    unsigned f1 = 0;
    if (nx.gather) {   // uniform
      tile_ctx_from_lds<false>(cn, t0 + nx.s * W, ldsMeta + nx.s * kMS);
      if (wave_u == 0) f1 = poll_issue(p.m, cn, p.m.flags);
    }
    dw_products(ldsX, ldsDZ, dwl, dbl);
    if (nx.gather) {
      if (wave_u == 0) {
        const bool hit = poll_ready(cn, f1, nx.ph);
        if (lane == 0) *s_pre = hit ? 1 : 0;
      }
      __syncthreads();
      pre = *s_pre != 0;
    } else {
      __syncthreads();
      pre = false;
    }
    const float4 gv = f4_sel(c.valid, f4_scale(c.ci, *reinterpret_cast<const float4 *>(&ldsG[c.grp * PG::TS + 4 * c.q])), f4_zero());
    if (pre) {   // uniform
      __syncthreads();   // every thread has read its row of G: the region is the halo again
      halo_fill_all(cn, nx.X, ldsXh);
    }
    if (c.valid) store_sc1(gout, own, gv);
    pend_flags = p.m.flags + 32 * c.tile;
    pend_ph = ph;
  };

  const bool multi = Kv > 1;   // (one tile: the next turn is this tile's own next phase, whose rows are stored in THIS turn)
  int ph = 0;
  {   // first phase: K-bar of the last stage of the last step = dt b_S lambda, layer 2's dense half (no gather, no wait)
    ++ph;
    const size_t ev = (size_t)((p.n_steps - 1) * S + (S - 1)) * 2 + 1;
    load_weight_lds(p.w2, ldsW, tid, false);
    if (Kv > 0) fetch(ev, 0);
    for (int s = 0; s < Kv; ++s) {
      TileCtx c;
      tile_ctx_from_lds<false>(c, t0 + s * W, ldsMeta + s * kMS);
      const unsigned own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
      t0_publish();   // (the phase's W is in LDS; the previous turn's products are done with the operand tiles)
      const float4 lam = f4_sel(c.valid, ld4_g(p.lam, own), f4_zero());
      const Aux mk = mk_n;
      const float4 xrow = xrow_n;
      const bool more = s + 1 < Kv;   // next: the workgroup's next tile, or layer 1 of the last stage of the last step on its first one
      Next nx;
      nx.gather = !more && multi; nx.ph = ph + 1; nx.s = more ? s + 1 : 0; nx.X = p.g2; nx.tape = true;
      nx.ev = more ? ev : (size_t)((p.n_steps - 1) * S + (S - 1)) * 2;
      dense(c, own, ph, dw2, db2, f4_scale(ldsC[S - 1], lam), mk, xrow, p.g2, nx);
    }
  }
  for (int n = p.n_steps - 1; n >= 0 && ok; --n) {
    for (int i = S - 1; i >= 0 && ok; --i) {
      const bool last = (i == 0 && n == 0);
      const size_t ev_b = (i >= 1) ? (size_t)(n * S + i - 1) * 2 + 1 : (size_t)((max(n, 1) - 1) * S + (S - 1)) * 2 + 1;   // layer-2 event behind this stage's layer 1
      {   // layer 1 of stage i: dL/dy1 = A^T g2
        ++ph;
        const size_t ev = (size_t)(n * S + i) * 2;
        __syncthreads();
        load_weight_lds(p.w1, ldsW, tid, false);
        for (int s = 0; s < Kv; ++s) {
          TileCtx c;
          tile_ctx_from_lds<false>(c, t0 + s * W, ldsMeta + s * kMS);
          const unsigned own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
          const Aux mk = mk_n;
          const float4 xrow = xrow_n;
          if (!top(c, ph, p.g2)) { ok = false; break; }
          const float4 t = tile_aggregate_lean(c, ldsXh);
          const bool more = s + 1 < Kv;
          Next nx;
          nx.gather = multi; nx.ph = more ? ph : ph + 1; nx.s = more ? s + 1 : 0; nx.X = more ? p.g2 : p.g1;
          nx.tape = more || !last; nx.ev = more ? ev : ev_b;
          dense(c, own, ph, dw1, db1, t, mk, xrow, p.g1, nx);
        }
        if (!ok) break;
      }
      {   // U-bar_i = A^T g1; K-bar of the stage evaluated before it (or the lambda update), layer 2's dense half
        ++ph;
        const size_t ev = ev_b;
        __syncthreads();
        load_weight_lds(p.w2, ldsW, tid, false);
        for (int s = 0; s < Kv; ++s) {
          TileCtx c;
          tile_ctx_from_lds<false>(c, t0 + s * W, ldsMeta + s * kMS);
          const unsigned own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
          const Aux mk = mk_n;        // (the last phase runs no dense half: nothing was asked for)
          const float4 xrow = xrow_n;
          if (!top(c, ph, p.g1)) { ok = false; break; }
          // lambda and the stage adjoints of this step from memory (rows 0..4 of ubar = U-bar_1..5): in flight under the aggregation
          const float4 lam = f4_sel(c.valid, ld4_g(p.lam, own), f4_zero());
          // (only the stage adjoints this stage combines, U-bar_j with i < j < S: the others meet a zero coefficient or are zero)
          const float4 lb1 = (i < 1 && S > 1) ? ld4_g(p.ubar, own) : f4_zero(), lb2 = (i < 2 && S > 2) ? ld4_g(p.ubar, own + rowb) : f4_zero(),
                       lb3 = (i < 3 && S > 3) ? ld4_g(p.ubar, own + 2 * rowb) : f4_zero(), lb4 = (i < 4 && S > 4) ? ld4_g(p.ubar, own + 3 * rowb) : f4_zero(),
                       lb5 = (i < 5 && S > 5) ? ld4_g(p.ubar, own + 4 * rowb) : f4_zero();
          const float4 t = tile_aggregate_lean(c, ldsXh);
          const float4 ub1 = i == 1 ? t : lb1, ub2 = i == 2 ? t : lb2, ub3 = i == 3 ? t : lb3, ub4 = i == 4 ? t : lb4,
                       ub5 = i == 5 ? t : lb5;   // U-bar_i is t itself
          float4 kbar;
          if (i >= 1) {
            if (c.valid) st4_g(p.ubar, own + (unsigned)(i - 1) * rowb, t);
            float4 v = f4_scale(ldsC[42 + i], t);
            v = f4_fma(ldsC[i - 1], lam, v);
            v = f4_fma(ldsC[6 + i * 6 + 2], ub2, v); v = f4_fma(ldsC[6 + i * 6 + 3], ub3, v);
            v = f4_fma(ldsC[6 + i * 6 + 4], ub4, v); v = f4_fma(ldsC[6 + i * 6 + 5], ub5, v);
            kbar = v;
          } else {
            float4 v = f4_scale(1.0f, t);
            v = f4_fma(1.0f, lam, v);
            v = f4_fma(1.0f, ub1, v); v = f4_fma(1.0f, ub2, v); v = f4_fma(1.0f, ub3, v);
            v = f4_fma(1.0f, ub4, v); v = f4_fma(1.0f, ub5, v);
            if (c.valid) st4_g(p.lam, own, v);
            kbar = f4_scale(ldsC[S - 1], v);
          }
          if (!last) {   // next: the workgroup's next tile, or layer 1 of stage i - 1 / of the last stage of the step before
            const bool more = s + 1 < Kv;
            const size_t ev_a = (size_t)(i >= 1 ? n * S + i - 1 : (max(n, 1) - 1) * S + (S - 1)) * 2;
            Next nx;
            nx.gather = multi; nx.ph = more ? ph : ph + 1; nx.s = more ? s + 1 : 0; nx.X = more ? p.g1 : p.g2;
            nx.tape = true; nx.ev = more ? ev : ev_a;
            dense(c, own, ph, dw2, db2, kbar, mk, xrow, p.g2, nx);
          } else {
            pre = false;     // (the final phase: every turn takes the blocking path; nothing is published)
            __syncthreads();
          }
        }
        if (!ok) break;
      }
    }
  }
  if (!ok) {
    __syncthreads();
    for (int s = 0; s < Kv; ++s) {
      const int4 sc = p.m.sched[(size_t)(t0 + s * W) * kTM + (tid >> 4)];
      if (sc.x >= 0) st4_g(p.lam, (unsigned)sc.x * (unsigned)(PD * 4) + (unsigned)((tid & 15) * 16), f4_nan());
    }
  }
  const float bad = __int_as_float(0x7fc00000);
  auto write_slab = [&](const f32x4 (&dwl)[PG::DWT], float dbl, float *slab_dw, float *slab_db) {
    float4 *slab4 = reinterpret_cast<float4 *>(slab_dw + (size_t)blockIdx.x * PD * PD);
#pragma unroll
    for (int mm = 0; mm < PG::DWT; ++mm) {
      const int tt = wave_u + PG::WAVES * mm;
      if (tt < NT) slab4[tt * 64 + lane] = f4_sel(ok, make_float4(dwl[mm][0], dwl[mm][1], dwl[mm][2], dwl[mm][3]), f4_nan());
    }
    if (dbpart == 0) slab_db[(size_t)blockIdx.x * PD + dbc] = ok ? dbl : bad;
  };
  write_slab(dw1, db1, p.slab_dw1, p.slab_db1);
  write_slab(dw2, db2, p.slab_dw2, p.slab_db2);
  if (tid == 0 && p.m.stats && Kv > 0) p.m.stats[2 * t0 + 1] = n_ahead;
}

}  // namespace

// ---- host side -----------------------------------------------------------------------------------------------------------

bool node_persistent_disabled_env() {   // read at every plan creation: tests switch it per plan
  const char *e = std::getenv("NGPDE_NO_PERSISTENT");
  return e && e[0] == '1';
}

// Wait lists: tile T waits for every tile that owns a row of T's halo in either direction, and for every tile whose halo holds
// a row of T (they read what T is about to overwrite).  [n_tiles][kNbrStride], -1 padded.  Returns false when a list overflows.
static bool build_wait_lists(const ngpde_graph *g, std::vector<int> &out) {
  const int nt = g->n_sched / kTileRows;
  std::vector<int32_t> order((size_t)g->n_nodes);
  if (!g->h_order.empty()) order = g->h_order;
  else if (hipMemcpy(order.data(), g->order, order.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
  std::vector<int32_t> tile_of((size_t)g->n_nodes, 0);
  for (int64_t pos = 0; pos < g->n_nodes; ++pos) tile_of[order[pos]] = (int32_t)(pos / kTileRows);
  std::vector<std::vector<int>> nb(nt);
  for (const Csr *c : {&g->by_t, &g->by_s}) {
    std::vector<int2> halo((size_t)nt * kHaloCap), info(nt);
    if (hipMemcpy(halo.data(), c->halo, halo.size() * sizeof(int2), hipMemcpyDeviceToHost) != hipSuccess) return false;
    if (hipMemcpy(info.data(), c->tile_info, info.size() * sizeof(int2), hipMemcpyDeviceToHost) != hipSuccess) return false;
    for (int t = 0; t < nt; ++t)
      for (int k = kTileRows; k < info[t].x; ++k) {
        const int u = tile_of[halo[(size_t)t * kHaloCap + k].x];
        if (u == t) continue;
        nb[t].push_back(u);
        nb[u].push_back(t);
      }
  }
  out.assign((size_t)nt * kNbrStride, -1);
  for (int t = 0; t < nt; ++t) {
    std::sort(nb[t].begin(), nb[t].end());
    nb[t].erase(std::unique(nb[t].begin(), nb[t].end()), nb[t].end());
    if ((int)nb[t].size() > kNbrStride - 1) return false;
    for (size_t k = 0; k < nb[t].size(); ++k) out[(size_t)t * kNbrStride + k] = nb[t][k];
  }
  return true;
}

// Can the plan run as two persistent launches?  (same conditions as the pre-scaled replayed plan, plus: d = 64, unweighted, ALL
// workgroups co-resident.)  0: no.  1: one tile per workgroup (graphs of at most CUs x occupancy tiles).  2: two tiles per
// workgroup through the two-slot kernels (up to twice as many tiles; relu when a backward is asked for).
int node_persistent_mode(const ngpde_graph *g, int d, int act, bool with_bwd) {
  if (node_persistent_disabled_env()) return 0;
  if (!g || d != PD || !fused_prescaled_supported(g, d)) return 0;
  const bool weighted = g->by_t.slot_w || g->by_s.slot_w;
  if (weighted && !(g->by_t.slot_w && g->by_s.slot_w)) return 0;
  int dev = 0, cus = 0, occ_f = 0, occ_b = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  // co-residency of the spin-waiting workgroups: the minimum over EVERY instantiation a plan of this kind can launch
  occ_f = occ_b = 1 << 30;
  auto take = [&](int &acc, auto kernel) {
    int o = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kernel, kThreads, 0) != hipSuccess) o = 0;
    acc = std::min(acc, o);
  };
  take(occ_f, node_fwd_persistent_kernel<NGPDE_ACT_RELU, true>);
  take(occ_f, node_fwd_persistent_kernel<NGPDE_ACT_RELU, false>);
  take(occ_f, node_fwd_persistent_kernel<-1, false>);
  take(occ_f, node_fwd_persistent2_kernel<NGPDE_ACT_RELU, true, false>);
  take(occ_f, node_fwd_persistent2_kernel<NGPDE_ACT_RELU, false, false>);
  take(occ_f, node_fwd_persistent2_kernel<-1, false, false>);
  take(occ_f, node_fwd_persistent2_kernel<NGPDE_ACT_RELU, true, true>);
  take(occ_f, node_fwd_persistent2_kernel<NGPDE_ACT_RELU, false, true>);
  take(occ_f, node_fwd_persistent2_kernel<-1, false, true>);
  take(occ_b, node_bwd_persistent_kernel<NGPDE_ACT_RELU>);
  take(occ_b, node_bwd_persistent_kernel<-1>);
  take(occ_f, node_fwd_persistent_kernel<-1, true>);
  take(occ_b, node_bwd_persistent2_kernel<false>);
  take(occ_b, node_bwd_persistent2_kernel<true>);
  const int nt = g->n_sched / kTileRows, resident = cus * std::min(occ_f, occ_b);
  if (nt < 1) return 0;
  if (weighted) {
    // edge weights: one tile per workgroup on the WGT one-tile kernels (slot weights in the LDS of one W; W1 as register fragments)
    // where the graph is one wave of workgroups, else the tile-round kernels (they keep one W in LDS anyway)
    int ow_f = 1 << 30, ow_b = 1 << 30;
    take(ow_f, node_fwd_persistent_kernel<NGPDE_ACT_RELU, true, true>);
    take(ow_f, node_fwd_persistent_kernel<NGPDE_ACT_RELU, false, true>);
    take(ow_f, node_fwd_persistent_kernel<-1, true, true>);
    take(ow_f, node_fwd_persistent_kernel<-1, false, true>);
    take(ow_b, node_bwd_persistent_kernel<NGPDE_ACT_RELU, true>);
    take(ow_b, node_bwd_persistent_kernel<-1, true>);
    const char *no1 = std::getenv("NGPDE_WEIGHTED_TILE_ROUNDS");   // 1: the tile-round kernels also where one tile per workgroup would do (A/B runs)
    if (nt <= cus * std::min(ow_f, ow_b) && !(no1 && no1[0] == '1')) return 1;
    return nt <= kMaxTileRoundsW * resident ? 3 : 0;
  }
  if (nt <= resident) return 1;
  const char *no_pairs = std::getenv("NGPDE_NO_TILE_PAIRS");
  if (no_pairs && no_pairs[0] == '1') return 0;
  const char *rounds = std::getenv("NGPDE_TILE_ROUNDS");   // 1: tile rounds also where tile pairs would do (A/B runs)
  if (nt <= 2 * resident && (!with_bwd || act == NGPDE_ACT_RELU) && !(rounds && rounds[0] == '1')) return 2;
  if (nt <= kMaxTileRounds * resident) return 3;   // K tiles per workgroup taking turns (node_*_persistentK_kernel)
  return 0;
}
int node_persistent_rounds(const ngpde_graph *g) {   // K of mode 3: tiles per workgroup
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  int occ = 1 << 30;
  auto take = [&](auto kernel) {
    int o = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kernel, kThreads, 0) != hipSuccess) o = 0;
    occ = std::min(occ, o);
  };
  take(node_fwd_persistentK_kernel<NGPDE_ACT_RELU, true>); take(node_fwd_persistentK_kernel<NGPDE_ACT_RELU, false>);
  take(node_fwd_persistentK_kernel<-1, true>); take(node_fwd_persistentK_kernel<-1, false>);
  take(node_fwd_persistentKP_kernel<NGPDE_ACT_RELU, true>); take(node_fwd_persistentKP_kernel<NGPDE_ACT_RELU, false>);
  take(node_fwd_persistentKP_kernel<-1, true>); take(node_fwd_persistentKP_kernel<-1, false>);
  take(node_bwd_persistentK_kernel<NGPDE_ACT_RELU>); take(node_bwd_persistentK_kernel<-1>);
  take(node_bwd_persistentKP_kernel<NGPDE_ACT_RELU>); take(node_bwd_persistentKP_kernel<-1>);
  if (g->by_t.slot_w) {
    take(node_fwd_persistentK_kernel<NGPDE_ACT_RELU, true, true>); take(node_fwd_persistentK_kernel<NGPDE_ACT_RELU, false, true>);
    take(node_fwd_persistentK_kernel<-1, true, true>); take(node_fwd_persistentK_kernel<-1, false, true>);
    take(node_bwd_persistentK_kernel<NGPDE_ACT_RELU, true>); take(node_bwd_persistentK_kernel<-1, true>);
  }
  const int nt = g->n_sched / kTileRows, resident = cus * occ;
  if (resident < 1) return 0;
  const int k = (nt + resident - 1) / resident;
  // (the K kernels' occupancy can be lower than what node_persistent_mode assumed from the one-tile kernels: a k beyond what the
  // kernels' LDS tables hold would leave tiles unprocessed -- their neighbours would spin into the timeout -- so it is refused here
  // and node_create keeps the replayed plan)
  if (k > (g->by_t.slot_w ? kMaxTileRoundsW : kMaxTileRounds)) return 0;
  return k;
}
bool node_persistent_supported(const ngpde_graph *g, int d, int act, bool with_bwd) { return node_persistent_mode(g, d, act, with_bwd) == 1; }

bool node_persistent_hub_possible(const ngpde_graph *g, int d) {
  const char *nh = std::getenv("NGPDE_NO_HALO");   // (asks for the per-row global gather everywhere)
  if (node_persistent_disabled_env() || (nh && nh[0] == '1')) return false;
  if (!g || d != PD || !g->has_norm || !g->self_loops || ((g->by_t.slot_w || g->by_s.slot_w) && !g->w_coo)) return false;
  if (g->by_t.halo_ok && g->by_s.halo_ok) return false;   // the 96-row geometry takes it
  int dev = 0, cus = 0, occ = 1 << 30;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  auto take = [&](auto kernel) {
    int o = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kernel, kThreads, 0) != hipSuccess) o = 0;
    occ = std::min(occ, o);
  };
  take(node_fwd_persistent_kernel<NGPDE_ACT_RELU, true, false, true>); take(node_fwd_persistent_kernel<NGPDE_ACT_RELU, false, false, true>);
  take(node_fwd_persistent_kernel<-1, true, false, true>); take(node_fwd_persistent_kernel<-1, false, false, true>);
  take(node_bwd_persistent_kernel<NGPDE_ACT_RELU, false, true>); take(node_bwd_persistent_kernel<-1, false, true>);
  const int nt = g->n_sched / kTileRows;
  return nt >= 1 && nt <= cus * occ;
}

// The hub geometry's lists of one direction, from the CSR lists and the schedule order (host copies, or downloaded from a
// device-built handle).  Slot numbering: own rows first (slot k = k-th row of the tile), every other row in order of first reference
// (rows of the tile in order, entries in CSR order).  false: some tile exceeds a cap.
// The hub geometry's own tile partition.  The handle's locality order grows clusters breadth-first, which puts a hub and the hubs
// next to it into ONE tile (Cora-shaped graphs: > 256 distinct rows).  Here the nodes are dealt out worst first, in order of
// descending degree: the n_tiles highest-degree nodes one per tile, every further node to the tile whose set of referenced rows H_t
// (members and all their neighbours, both directions) grows least by it, among the tiles with a free slot that stay within the cap
// with one row reserved per slot still free; a node with more than kSlotWidth entries (a long row: summed by the whole workgroup, two
// barriers each) prefers the tiles with the fewest long rows.  Leaves end up with the hub they hang on (growth 0), hubs apart from each
// other: at BASELINE config 1's shape one long row per tile at most, 94 referenced rows per tile on average, 131 at most (the first form
// of this deal -- least growth only -- put the hubs next to each other: the tile with most of them took 17 k cycles per phase and every
// other tile waited for it; the second -- hubs apart, least growth -- filled the largest hub's tile to the cap of 255 rows).  Returns the positions -> node order (tile t = positions 32 t ..), or an empty vector when some node fits nowhere.
static std::vector<int32_t> hub_partition(int64_t n, const std::vector<int32_t> rp[2], const std::vector<int32_t> cl[2], bool balance) {
  const int nt = (int)((n + kTileRows - 1) / kTileRows);
  std::vector<std::vector<int32_t>> nb((size_t)n);
  for (int64_t v = 0; v < n; ++v) {
    std::vector<int32_t> &a = nb[(size_t)v];
    for (int dir = 0; dir < 2; ++dir)
      for (int32_t p = rp[dir][v]; p < rp[dir][v + 1]; ++p)
        if (cl[dir][p] != v) a.push_back(cl[dir][p]);
    std::sort(a.begin(), a.end());
    a.erase(std::unique(a.begin(), a.end()), a.end());
  }
  std::vector<int32_t> by_degree((size_t)n);
  for (int64_t v = 0; v < n; ++v) by_degree[(size_t)v] = (int32_t)v;
  std::stable_sort(by_degree.begin(), by_degree.end(), [&](int32_t a, int32_t b) { return nb[a].size() > nb[b].size(); });
  std::vector<uint8_t> in_h((size_t)nt * n, 0);
  std::vector<int> h_size(nt, 0), cap(nt, kTileRows);
  std::vector<int> n_long(nt, 0);
  std::vector<std::vector<int32_t>> members(nt);
  cap[nt - 1] = (int)(n - (int64_t)kTileRows * (nt - 1));
  int64_t dealt = 0;
  for (int32_t v : by_degree) {
    const bool is_long = (int)nb[v].size() > kSlotWidth;
    int best_t = -1, best_score = 1 << 30, best_l = 1 << 30;
    if (dealt < nt) best_t = (int)dealt;   // the n_tiles highest-degree nodes: one per tile
    ++dealt;
    for (int t = 0; t < nt && dealt > nt; ++t) {
      if ((int)members[t].size() >= cap[t]) continue;
      const uint8_t *h = in_h.data() + (size_t)t * n;
      int inc = h[v] ? 0 : 1;
      for (int32_t w : nb[v]) inc += h[w] ? 0 : 1;
      if (h_size[t] + inc + (cap[t] - (int)members[t].size() - 1) > kHubHalo) continue;
      const int l = is_long ? n_long[t] : 0;
      // growth, plus a sixteenth of what the tile already references: every tile waits for the slowest one each phase, so a node that
      // would add one row to a tile of 200 goes to a tile of 60 even if it adds three there (largest tile at config 1's shape: 255 -> 131
      // rows, the average unchanged at 94)
      const int score = balance ? 16 * inc + h_size[t] : 512 * inc + h_size[t];   // (not balanced: least growth, then the smaller tile)
      if (l < best_l || (l == best_l && score < best_score)) { best_t = t; best_score = score; best_l = l; }
    }
    if (best_t < 0) return {};
    uint8_t *h = in_h.data() + (size_t)best_t * n;
    if (!h[v]) { h[v] = 1; ++h_size[best_t]; }
    for (int32_t w : nb[v])
      if (!h[w]) { h[w] = 1; ++h_size[best_t]; }
    if (h_size[best_t] + (cap[best_t] - (int)members[best_t].size() - 1) > kHubHalo) return {};   // (a seed beyond the cap: a hub of > ~224 neighbours)
    members[best_t].push_back(v);
    n_long[best_t] += is_long ? 1 : 0;
  }
  std::vector<int32_t> order;
  order.reserve((size_t)n);
  for (int t = 0; t < nt; ++t) order.insert(order.end(), members[t].begin(), members[t].end());
  return order;
}

struct HubHost {
  std::vector<int32_t> halo;
  std::vector<float> w;   // beside `slots` (weighted graphs), else empty
  std::vector<uint8_t> slots, longs;
  std::vector<int2> rows;
  std::vector<int4> info;
};
static bool host_csr(const ngpde_graph *g, const Csr &c, std::vector<int32_t> &rowptr, std::vector<int32_t> &col) {
  const int64_t n = g->n_nodes, m = g->n_edges;
  if (!c.h_rowptr.empty()) { rowptr = c.h_rowptr; col = c.h_col; return true; }
  rowptr.resize((size_t)n + 1); col.resize((size_t)std::max<int64_t>(m, 1));
  if (hipMemcpy(rowptr.data(), c.rowptr, rowptr.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
  if (m > 0 && hipMemcpy(col.data(), c.col, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
  return true;
}
// w_csr: the entries' edge weights in the list's own order, or NULL
static bool build_hub_lists(int64_t n, int nt, const std::vector<int32_t> &rowptr, const std::vector<int32_t> &col,
                            const std::vector<int32_t> &order, HubHost &o, const float *w_csr = nullptr) {
  o.halo.assign((size_t)nt * kHubHalo, 0);
  o.slots.assign((size_t)nt * kHubList, 0);
  if (w_csr) o.w.assign((size_t)nt * kHubList, 0.f);
  o.longs.assign((size_t)nt * kTileRows, 0);
  o.rows.assign((size_t)nt * kTileRows, make_int2(0, 0));
  o.info.assign((size_t)nt, make_int4(0, 0, 0, 0));
  std::vector<int32_t> slot_of((size_t)n, -1), stamp((size_t)n, -1);
  for (int t = 0; t < nt; ++t) {
    int count = kTileRows, bytes = 0, n_long = 0;
    for (int k = 0; k < kTileRows; ++k) {
      const int64_t pos = (int64_t)t * kTileRows + k;
      if (pos >= n) continue;
      const int32_t v = order[pos];
      stamp[v] = t; slot_of[v] = k;
      o.halo[(size_t)t * kHubHalo + k] = v;
    }
    for (int k = 0; k < kTileRows; ++k) {
      const int64_t pos = (int64_t)t * kTileRows + k;
      if (pos >= n) break;
      const int32_t v = order[pos];
      const int32_t rs = rowptr[v], deg = rowptr[v + 1] - rs;
      if (bytes + deg > kHubList) return false;
      o.rows[(size_t)pos] = make_int2(bytes, deg);
      if (deg > kSlotWidth) o.longs[(size_t)t * kTileRows + n_long++] = (uint8_t)k;
      for (int j = 0; j < deg; ++j) {
        const int32_t u = col[rs + j];
        if (stamp[u] != t) {
          if (count >= kHubHalo) return false;
          stamp[u] = t; slot_of[u] = count;
          o.halo[(size_t)t * kHubHalo + count++] = u;
        }
        o.slots[(size_t)t * kHubList + bytes + j] = (uint8_t)slot_of[u];
        if (w_csr) o.w[(size_t)t * kHubList + bytes + j] = w_csr[rs + j];
      }
      bytes = (bytes + deg + 3) & ~3;
      if (bytes > kHubList) return false;
    }
    o.info[t] = make_int4(count, (bytes + 15) & ~15, n_long, 0);
  }
  return true;
}

// Host only (no device call): the hub geometry's tile partition of a graph given as its two CSR lists -- what node_persistent_setup
// would use -- so that a caller (and the CPU tests, tests/test_hub_partition.py) can ask whether a graph with hubs fits the persistent
// solver.  order[32 t + k] = the node in row k of tile t; tile_rows[2 t + dir] = the distinct rows tile t references in direction
// dir (0: lists by target, 1: by source; members included).
}  // namespace ngpde
extern "C" int32_t ngpde_hub_partition_host(int64_t n_nodes, const int32_t *rowptr_by_target, const int32_t *col_by_target,
                                            const int32_t *rowptr_by_source, const int32_t *col_by_source, int32_t *order,
                                            int32_t *tile_rows) {
  using namespace ngpde;
  NGPDE_REQUIRE(n_nodes > 0 && rowptr_by_target && rowptr_by_source && order, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_hub_partition_host: n_nodes > 0, both row pointer arrays and `order` are required");
  const int nt = (int)((n_nodes + kTileRows - 1) / kTileRows);
  NGPDE_REQUIRE(nt <= kHubNbr, NGPDE_ERR_UNSUPPORTED, "persistent solver, hub geometry: at most %d tiles, one per workgroup", kHubNbr);
  std::vector<int32_t> rp[2], cl[2];
  const int32_t *rps[2] = {rowptr_by_target, rowptr_by_source}, *cls[2] = {col_by_target, col_by_source};
  for (int dir = 0; dir < 2; ++dir) {
    rp[dir].assign(rps[dir], rps[dir] + n_nodes + 1);
    const int64_t m = rp[dir][(size_t)n_nodes];
    NGPDE_REQUIRE(rp[dir][0] == 0 && m >= 0 && (m == 0 || cls[dir]), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_hub_partition_host: CSR list %d is malformed", dir);
    for (int64_t v = 0; v < n_nodes; ++v)
      NGPDE_REQUIRE(rp[dir][v] <= rp[dir][v + 1], NGPDE_ERR_INVALID_ARGUMENT, "ngpde_hub_partition_host: row pointers of list %d decrease at row %lld", dir, (long long)v);
    cl[dir].assign(cls[dir], cls[dir] + m);
    for (int64_t e = 0; e < m; ++e)
      NGPDE_REQUIRE(cl[dir][e] >= 0 && cl[dir][e] < n_nodes, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_hub_partition_host: list %d names node %d of %lld", dir, cl[dir][e], (long long)n_nodes);
  }
  std::vector<int32_t> ord = hub_partition(n_nodes, rp, cl, true);
  if ((int64_t)ord.size() != n_nodes) ord = hub_partition(n_nodes, rp, cl, false);
  NGPDE_REQUIRE((int64_t)ord.size() == n_nodes, NGPDE_ERR_UNSUPPORTED,
                "persistent solver, hub geometry: no partition into 32-row tiles of at most %d referenced rows each (a node of more than %d distinct in+out neighbours, or tiles that do not close)",
                kHubHalo, kHubHalo - kTileRows);
  HubHost hh[2];
  NGPDE_REQUIRE(build_hub_lists(n_nodes, nt, rp[0], cl[0], ord, hh[0]) && build_hub_lists(n_nodes, nt, rp[1], cl[1], ord, hh[1]), NGPDE_ERR_UNSUPPORTED,
                "persistent solver, hub geometry: a tile references more than %d distinct rows or holds more than %d entries", kHubHalo, kHubList);
  std::copy(ord.begin(), ord.end(), order);
  if (tile_rows)
    for (int t = 0; t < nt; ++t)
      for (int dir = 0; dir < 2; ++dir) tile_rows[2 * t + dir] = hh[dir].info[(size_t)t].x;
  return NGPDE_OK;
}
namespace ngpde {

// NGPDE_NO_INTERLEAVE=1: a batch's members one after the other (the round-2 form) instead of two at a time -- the A/B switch and
// the reference the interleaved kernels are compared with bit for bit
bool node_persistent_interleave_env() {
  const char *e = std::getenv("NGPDE_NO_INTERLEAVE");
  return !(e && e[0] == '1');
}

// ---- a plan's own-first slot tables (OwnFirst, common.h) ----------------------------------------------------------------------------
namespace {
// one 64-thread workgroup per tile, thread r < 32 = row r: the row's slot bytes with the own-tile slots (< 32) first, padded with the
// all-zero row (kHaloCap) to 4 x the own rounds of the row's group of four (= a wave of the 64-wide kernels), then the foreign slots;
// weights travel with their slots; a group in which some row would not fit its 32 bytes keeps the handle's order (pre = 0)
__global__ __launch_bounds__(64) void own_first_tables_kernel(int n_tiles, const uint8_t *__restrict__ slots, const int4 *__restrict__ sched,
                                                              const float *__restrict__ slot_w, uint8_t *__restrict__ o_slots,
                                                              int4 *__restrict__ o_sched, float *__restrict__ o_w, uint8_t *__restrict__ o_pre) {
  const int tile = blockIdx.x, r = threadIdx.x;
  const bool row = r < kTileRows;
  const size_t pos = (size_t)tile * kTileRows + (row ? r : 0);
  int4 sc = sched[pos];
  const int deg = (row && sc.x >= 0) ? min(sc.z, kSlotWidth) : 0;
  uint8_t b[kSlotWidth];
  {
    const uint4 lo = reinterpret_cast<const uint4 *>(slots + pos * kSlotWidth)[0], hi = reinterpret_cast<const uint4 *>(slots + pos * kSlotWidth)[1];
    const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int j = 0; j < kSlotWidth; ++j) b[j] = (uint8_t)((w[j >> 2] >> (8 * (j & 3))) & 0xff);
  }
  int no = 0, nf = 0;
#pragma unroll
  for (int j = 0; j < kSlotWidth; ++j) {
    if (j < deg) {
      if (b[j] < kTileRows) ++no;
      else ++nf;
    }
  }
  int pre = (no + 3) >> 2;
  pre = max(pre, __shfl_xor(pre, 1));
  pre = max(pre, __shfl_xor(pre, 2));
  int fits = (4 * pre + nf <= kSlotWidth) ? 1 : 0;
  fits = min(fits, __shfl_xor(fits, 1));
  fits = min(fits, __shfl_xor(fits, 2));
  if (!fits) pre = 0;
  if (row) {
    uint8_t *ob = o_slots + pos * kSlotWidth;
    float *ow = o_w ? o_w + pos * kSlotWidth : nullptr;
    const float *iw = slot_w ? slot_w + pos * kSlotWidth : nullptr;
    if (fits) {
      int k = 0;
      for (int j = 0; j < deg; ++j)
        if (b[j] < kTileRows) { ob[k] = b[j]; if (ow) ow[k] = iw[j]; ++k; }
      for (; k < 4 * pre; ++k) { ob[k] = (uint8_t)kHaloCap; if (ow) ow[k] = 0.f; }
      for (int j = 0; j < deg; ++j)
        if (b[j] >= kTileRows) { ob[k] = b[j]; if (ow) ow[k] = iw[j]; ++k; }
      const int len = k;
      for (; k < kSlotWidth; ++k) { ob[k] = (uint8_t)kHaloCap; if (ow) ow[k] = 0.f; }
      if (sc.x >= 0) sc.z = len;
    } else {
      for (int j = 0; j < kSlotWidth; ++j) { ob[j] = b[j]; if (ow) ow[j] = iw[j]; }
    }
    o_sched[pos] = sc;
    if ((r & 3) == 0) o_pre[(size_t)tile * 8 + (r >> 2)] = (uint8_t)pre;
  }
}
}  // namespace

int32_t own_first_tables_build(const ngpde_graph *g, OwnFirst *of, hipStream_t stream) {
  const int nt = g->n_sched / kTileRows;
  const Csr *cs[2] = {&g->by_t, &g->by_s};
  // the by-target lists only: the adjoint's tiles reach their wait with the flags already set (its parameter-gradient products sit
  // between publish and wait), so there the padding rounds cost more than the early sums win (2.913 -> 2.925 ms, profiles/r06_l_own_first.txt);
  // NGPDE_OWN_FIRST_ADJOINT=1 builds both (A/B runs)
  const char *adj = std::getenv("NGPDE_OWN_FIRST_ADJOINT");
  const int n_dir = (adj && adj[0] == '1') ? 2 : 1;
  for (int dir = 0; dir < n_dir; ++dir) {
    const Csr &c = *cs[dir];
    if (!c.halo_ok || !c.slots || !c.sched) continue;
    NGPDE_HIP_CHECK(hipMalloc((void **)&of->slots[dir], (size_t)g->n_sched * kSlotWidth));
    NGPDE_HIP_CHECK(hipMalloc((void **)&of->sched[dir], (size_t)g->n_sched * sizeof(int4)));
    NGPDE_HIP_CHECK(hipMalloc((void **)&of->pre[dir], (size_t)nt * 8));
    if (c.slot_w) NGPDE_HIP_CHECK(hipMalloc((void **)&of->slot_w[dir], (size_t)g->n_sched * kSlotWidth * sizeof(float)));
    hipLaunchKernelGGL(own_first_tables_kernel, dim3(nt), dim3(64), 0, stream, nt, c.slots, c.sched, c.slot_w, of->slots[dir], of->sched[dir],
                       of->slot_w[dir], of->pre[dir]);
    NGPDE_LAUNCH_CHECK("own_first_tables_kernel");
  }
  NGPDE_HIP_CHECK(hipStreamSynchronize(stream));
  return NGPDE_OK;
}

void own_first_tables_free(OwnFirst *of) {
  for (int dir = 0; dir < 2; ++dir) {
    if (of->slots[dir]) (void)hipFree(of->slots[dir]);
    if (of->sched[dir]) (void)hipFree(of->sched[dir]);
    if (of->slot_w[dir]) (void)hipFree(of->slot_w[dir]);
    if (of->pre[dir]) (void)hipFree(of->pre[dir]);
  }
  *of = OwnFirst();
}

int32_t node_persistent_setup(const ngpde_graph *g, const float *coef_host, NodePersist *ps, bool pair, bool hub) {
  std::vector<int> lists;
  const int nt = g->n_sched / kTileRows;
  ps->hub = false;
  if (hub) {
    NGPDE_REQUIRE(!pair && nt <= kHubNbr, NGPDE_ERR_UNSUPPORTED, "persistent solver, hub geometry: at most %d tiles, one per workgroup", kHubNbr);
    std::vector<int32_t> rp[2], cl[2];
    NGPDE_REQUIRE(host_csr(g, g->by_t, rp[0], cl[0]) && host_csr(g, g->by_s, rp[1], cl[1]), NGPDE_ERR_HIP, "download of the CSR lists failed");
    // (the balanced deal can strand the last nodes when a hub's tile is near the cap and only its own leaves would fit it: then least growth alone)
    std::vector<int32_t> order = hub_partition(g->n_nodes, rp, cl, true);
    if ((int64_t)order.size() != g->n_nodes) order = hub_partition(g->n_nodes, rp, cl, false);
    NGPDE_REQUIRE((int64_t)order.size() == g->n_nodes, NGPDE_ERR_UNSUPPORTED,
                  "persistent solver, hub geometry: no partition into 32-row tiles of at most %d referenced rows each (a node of more than %d distinct in+out neighbours, or tiles that do not close)",
                  kHubHalo, kHubHalo - kTileRows);
    // edge weights: the handle's COO copy through each list's entry -> COO position map
    std::vector<float> wcsr[2];
    if (g->w_coo && g->n_edges > 0) {
      std::vector<float> wc((size_t)g->n_edges);
      NGPDE_HIP_CHECK(hipMemcpy(wc.data(), g->w_coo, wc.size() * sizeof(float), hipMemcpyDeviceToHost));
      const Csr *cs[2] = {&g->by_t, &g->by_s};
      for (int dir = 0; dir < 2; ++dir) {
        std::vector<int32_t> eid = cs[dir]->h_eid;
        if (eid.empty()) {
          eid.resize((size_t)g->n_edges);
          NGPDE_HIP_CHECK(hipMemcpy(eid.data(), cs[dir]->eid, eid.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        }
        wcsr[dir].resize((size_t)g->n_edges);
        for (int64_t e = 0; e < g->n_edges; ++e) wcsr[dir][(size_t)e] = wc[(size_t)eid[(size_t)e]];
      }
    }
    HubHost hh[2];
    NGPDE_REQUIRE(build_hub_lists(g->n_nodes, nt, rp[0], cl[0], order, hh[0], wcsr[0].empty() ? nullptr : wcsr[0].data()) &&
                      build_hub_lists(g->n_nodes, nt, rp[1], cl[1], order, hh[1], wcsr[1].empty() ? nullptr : wcsr[1].data()),
                  NGPDE_ERR_UNSUPPORTED,
                  "persistent solver, hub geometry: a tile references more than %d distinct rows or holds more than %d entries", kHubHalo, kHubList);
    // the partition's schedule: {node, 0, 0, bits of c[node]} per position, -1 padded
    std::vector<float> cnode((size_t)g->n_nodes);
    NGPDE_HIP_CHECK(hipMemcpy(cnode.data(), g->c, cnode.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<int4> hsched((size_t)nt * kTileRows, make_int4(-1, 0, 0, 0));
    for (int64_t pos = 0; pos < g->n_nodes; ++pos) {
      int4 e = make_int4(order[pos], 0, 0, 0);
      std::memcpy(&e.w, &cnode[order[pos]], 4);
      hsched[(size_t)pos] = e;
    }
    // wait lists: symmetric closure over both directions, up to 255 tiles, position 63 left free (that lane watches the abort word)
    std::vector<int32_t> tile_of((size_t)g->n_nodes, 0);
    for (int64_t pos = 0; pos < g->n_nodes; ++pos) tile_of[order[pos]] = (int32_t)(pos / kTileRows);
    std::vector<std::vector<int>> nb(nt);
    for (const HubHost &h : hh)
      for (int t = 0; t < nt; ++t)
        for (int k = kTileRows; k < h.info[t].x; ++k) {
          const int u = tile_of[h.halo[(size_t)t * kHubHalo + k]];
          if (u == t) continue;
          nb[t].push_back(u);
          nb[u].push_back(t);
        }
    lists.assign((size_t)nt * kHubNbr, -1);
    for (int t = 0; t < nt; ++t) {
      std::sort(nb[t].begin(), nb[t].end());
      nb[t].erase(std::unique(nb[t].begin(), nb[t].end()), nb[t].end());
      NGPDE_REQUIRE((int)nb[t].size() <= kHubNbr - 1, NGPDE_ERR_UNSUPPORTED, "persistent solver, hub geometry: a wait list exceeds %d tiles", kHubNbr - 1);
      for (size_t k = 0; k < nb[t].size(); ++k) lists[(size_t)t * kHubNbr + (k < 63 ? k : k + 1)] = nb[t][k];
    }
    for (int dir = 0; dir < 2; ++dir) {
      NodePersist::HubLists &L = ps->hub_lists[dir];
      const HubHost &h = hh[dir];
      auto up = [&](auto **dst, const auto &v) -> int32_t {
        NGPDE_HIP_CHECK(hipMalloc((void **)dst, std::max<size_t>(v.size(), 1) * sizeof(v[0])));
        NGPDE_HIP_CHECK(hipMemcpy(*dst, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice));
        return NGPDE_OK;
      };
      int32_t st;
      if ((st = up(&L.halo, h.halo)) || (st = up(&L.slots, h.slots)) || (st = up(&L.rows, h.rows)) || (st = up(&L.info, h.info)) ||
          (st = up(&L.longs, h.longs)) || (st = up(&L.sched, hsched)))
        return st;
      if (!h.w.empty() && (st = up(&L.w, h.w))) return st;
    }
    ps->hub = true;
  } else {
    NGPDE_REQUIRE(build_wait_lists(g, lists), NGPDE_ERR_UNSUPPORTED, "persistent solver: a tile's wait list exceeds %d tiles", kNbrStride);
  }
  ps->n_tiles = nt;
  ps->pair_wgs = 0;
  if (pair) {
    // workgroup b holds tiles t = xcd_tile(b, W) and t + W (W = ceil(nt / 2)): half a schedule apart, so never neighbours on a
    // graph with any locality -- but it is checked: a tile waiting for its own workgroup's other tile would wait for a flag that is
    // published only after the wait
    const int W = (nt + 1) / 2;
    for (int t = 0; t + W < nt; ++t)
      for (int k = 0; k < kNbrStride; ++k)
        NGPDE_REQUIRE(lists[(size_t)t * kNbrStride + k] != t + W, NGPDE_ERR_UNSUPPORTED,
                      "persistent solver: tiles %d and %d of one workgroup are neighbours", t, t + W);
    ps->pair_wgs = W;
  }
  NGPDE_HIP_CHECK(hipMalloc((void **)&ps->nbr, lists.size() * sizeof(int)));
  NGPDE_HIP_CHECK(hipMemcpy(ps->nbr, lists.data(), lists.size() * sizeof(int), hipMemcpyHostToDevice));
  // flag words: one 128-byte line per tile and slot (two slots: the interleaved kernels), + one line for the abort word; zeroed
  // (by a kernel) before every launch
  ps->sync_bytes = (size_t)(2 * nt + 1) * 128;
  NGPDE_HIP_CHECK(hipMalloc((void **)&ps->sync, ps->sync_bytes));
  NGPDE_HIP_CHECK(hipMemset(ps->sync, 0, ps->sync_bytes));
  NGPDE_HIP_CHECK(hipMalloc((void **)&ps->coef, 90 * sizeof(float)));
  NGPDE_HIP_CHECK(hipMemcpy(ps->coef, coef_host, 90 * sizeof(float), hipMemcpyHostToDevice));
  // the sticky fault word lives in pinned, device-mapped HOST memory: the latch kernel writes it through the device pointer, and
  // every later entry of the plan can look at it without a synchronisation (ngpde_node_gcn2_forward / _backward refuse to go on)
  NGPDE_HIP_CHECK(hipHostMalloc((void **)&ps->fault_host, 128, hipHostMallocMapped));
  *ps->fault_host = 0u;
  NGPDE_HIP_CHECK(hipHostGetDevicePointer((void **)&ps->fault, const_cast<unsigned *>(ps->fault_host), 0));
  NGPDE_HIP_CHECK(hipMalloc((void **)&ps->stats, (size_t)nt * 2 * sizeof(int)));
  NGPDE_HIP_CHECK(hipMemset(ps->stats, 0, (size_t)nt * 2 * sizeof(int)));
  return NGPDE_OK;
}

void node_persistent_free(NodePersist *ps) {
  if (ps->nbr) (void)hipFree(ps->nbr);
  if (ps->sync) (void)hipFree(ps->sync);
  if (ps->fault_host) (void)hipHostFree(const_cast<unsigned *>(ps->fault_host));
  ps->fault_host = nullptr;
  if (ps->coef) (void)hipFree(ps->coef);
  if (ps->stats) (void)hipFree(ps->stats);
  ps->stats = nullptr;
  for (NodePersist::HubLists &L : ps->hub_lists) {
    if (L.halo) (void)hipFree(L.halo);
    if (L.slots) (void)hipFree(L.slots);
    if (L.rows) (void)hipFree(L.rows);
    if (L.info) (void)hipFree(L.info);
    if (L.longs) (void)hipFree(L.longs);
    if (L.sched) (void)hipFree(L.sched);
    if (L.w) (void)hipFree(L.w);
    L = NodePersist::HubLists();
  }
  ps->hub = false;
  ps->nbr = nullptr; ps->sync = nullptr; ps->fault = nullptr; ps->coef = nullptr;
}

namespace {
// fault |= abort word of the launch that just ran (sticky, read by ngpde_node_fault)
// NGPDE_DEBUG_FORCE_ABORT=1 (tests only): the launch starts with its abort word set, i.e. every workgroup gives up at its first wait
__global__ void set_word_kernel(unsigned *w, unsigned v) {
  if (threadIdx.x == 0) *w = v;
}
__global__ void latch_fault_kernel(const unsigned *abort_word, unsigned *fault) {
  if (threadIdx.x == 0 && *abort_word != 0) *fault = 1u;
}
}  // namespace
const unsigned *node_persistent_abort_word(const NodePersist *ps) { return ps->sync ? ps->sync + (size_t)ps->n_tiles * 64 : nullptr; }
namespace {
TileMeta make_meta(const Csr &c, const NodePersist &ps, int dir, const OwnFirst *of) {
  TileMeta m;
  m.halo = c.halo; m.slots = c.slots; m.sched = c.sched; m.tile_info = c.tile_info; m.nbr = ps.nbr; m.slot_w = c.slot_w;
  m.of_pre = nullptr;
  if (of && of->slots[dir] && !ps.hub) {   // the plan's own-first tables: the same rows in another order, padded lengths in the schedule
    m.slots = of->slots[dir]; m.sched = of->sched[dir]; m.of_pre = of->pre[dir];
    if (m.slot_w) m.slot_w = of->slot_w[dir];
  }
  const NodePersist::HubLists &L = ps.hub_lists[dir];
  m.hub_halo = L.halo; m.hub_slots = L.slots; m.hub_rows = L.rows; m.hub_info = L.info; m.hub_long = L.longs; m.hub_sched = L.sched;
  m.hub_w = L.w;
  if (ps.hub) m.slot_w = nullptr;   // (the hub geometry reads its own weight lists)
  m.flags = ps.sync; m.abort_word = ps.sync + (size_t)ps.n_tiles * 64; m.n_tiles = ps.n_tiles;   // [slot 0 | slot 1 | abort]
  m.stats = ps.stats;
#ifdef NGPDE_STAMPS
  m.stamps = g_pst_base; m.stamps_max = g_pst_max;
#endif
  return m;
}
}  // namespace

#ifdef NGPDE_STAMPS
extern "C" int32_t ngpde_debug_set_persistent_stamps(unsigned long long *dev_buf, int32_t max_phases) {
  g_pst_base = dev_buf;   // [n_tiles][max_phases][8], or NULL
  g_pst_max = max_phases;
  return NGPDE_OK;
}
#endif

// A persistent launch needs ALL its workgroups resident, and two of them in flight on one device (two plans on two streams) can
// starve each other of residency until both time out.  Inside one process they are therefore made to take turns: every persistent
// launch first makes its stream wait for the event recorded behind the previous persistent launch on that device, whatever stream
// that one ran on.  (Nothing can be done about another PROCESS on the same device: one rank per GPU, include/ngpde.h.)
namespace {
struct Turnstile {
  std::mutex mu;
  hipEvent_t last[16] = {};
  bool used[16] = {};
};
Turnstile &turnstile() {
  static Turnstile t;
  return t;
}
}  // namespace

int32_t PersistentTurn::enter(hipStream_t s) {
  stream = s;
  NGPDE_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return NGPDE_OK;
  Turnstile &t = turnstile();
  t.mu.lock();
  held = true;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) cs = hipStreamCaptureStatusNone;
  if (cs == hipStreamCaptureStatusNone && t.used[dev]) NGPDE_HIP_CHECK(hipStreamWaitEvent(stream, t.last[dev], 0));
  return NGPDE_OK;
}
int32_t PersistentTurn::leave() {
  if (!held) return NGPDE_OK;
  Turnstile &t = turnstile();
  int32_t st = NGPDE_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) cs = hipStreamCaptureStatusNone;
  if (cs == hipStreamCaptureStatusNone) {
    if (!t.last[dev] && hipEventCreateWithFlags(&t.last[dev], hipEventDisableTiming) != hipSuccess) st = fail(NGPDE_ERR_HIP, "hipEventCreate failed");
    if (st == NGPDE_OK && hipEventRecord(t.last[dev], stream) != hipSuccess) st = fail(NGPDE_ERR_HIP, "hipEventRecord failed");
    if (st == NGPDE_OK) t.used[dev] = true;
  }
  held = false;
  t.mu.unlock();
  return st;
}
PersistentTurn::~PersistentTurn() {
  if (held) {
    held = false;
    turnstile().mu.unlock();
  }
}

int32_t launch_node_fwd_persistent(const NodePersistFwd &a, hipStream_t stream) {
  const ngpde_graph *g = a.g;
  const NodePersist &ps = *a.ps;
  int32_t st;
  PersistentTurn turn;
  if ((st = turn.enter(stream))) return st;
  if ((st = launch_zero(ps.sync, ps.sync_bytes, stream))) return st;
  PFwdK k;
  k.m = make_meta(g->by_t, ps, 0, a.of);
  {
    const char *fa = std::getenv("NGPDE_DEBUG_FORCE_ABORT");
    if (fa && fa[0] == '1') hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(64), 0, stream, k.m.abort_word, 1u);
  }
  k.n_steps = a.n_steps; k.S = a.S; k.act = a.act; k.n_members = a.n_members;
  k.u_in = a.u_in; k.u_out = a.u_out; k.bufA = a.bufA; k.bufB = a.bufB;
  k.w1 = a.w1; k.b1 = a.b1; k.w2 = a.w2; k.b2 = a.b2;
  k.tape = a.tape; k.masks = a.masks; k.row_elems = a.row_elems; k.mask_bytes = a.mask_bytes;
  k.flag_stride = (size_t)ps.n_tiles * 32;
  k.pair_wgs = a.pair ? ps.pair_wgs : 0;
  k.cf = ps.coef;
  NGPDE_REQUIRE(!a.pair || (ps.pair_wgs > 0 && !a.interleave && a.n_members == 1), NGPDE_ERR_STATE, "tile-pair launch without its setup");
  k.k_tiles = a.k_tiles; k.state = a.state;
  if (a.k_tiles > 0) {   // tile rounds: K tiles per workgroup, state in memory
    NGPDE_REQUIRE(ps.pair_wgs > 0 && a.n_members == 1 && a.state, NGPDE_ERR_STATE, "tile-round launch without its setup");
    NGPDE_REQUIRE(a.k_tiles <= kMaxTileRounds, NGPDE_ERR_STATE, "tile rounds: at most %d tiles per workgroup", kMaxTileRounds);
    k.pair_wgs = ps.pair_wgs; k.ztape = a.ztape;
    if ((st = launch_zero(a.state + a.row_elems, 6 * a.row_elems * sizeof(float), stream))) return st;   // k_0 .. k_5 start from zero
    const dim3 gridk(ps.pair_wgs), blockk(kThreads);
    NGPDE_REQUIRE(!k.m.slot_w || a.k_tiles <= kMaxTileRoundsW, NGPDE_ERR_STATE, "weighted tile rounds: at most %d tiles per workgroup", kMaxTileRoundsW);
    const char *nopipe = std::getenv("NGPDE_NO_TILE_PIPE");
    const bool pipe = !k.m.slot_w && !(nopipe && nopipe[0] == '1');
#define NGPDE_PFK_LAUNCH(AA, TT)                                                                                                  \
    if (pipe) {                                                                                                                   \
      if (a.ev_start) hipExtLaunchKernelGGL((node_fwd_persistentKP_kernel<AA, TT>), gridk, blockk, 0, stream, a.ev_start, a.ev_stop, 0, k);  \
      else hipLaunchKernelGGL((node_fwd_persistentKP_kernel<AA, TT>), gridk, blockk, 0, stream, k);                                \
    } else if (k.m.slot_w) {                                                                                                             \
      if (a.ev_start) hipExtLaunchKernelGGL((node_fwd_persistentK_kernel<AA, TT, true>), gridk, blockk, 0, stream, a.ev_start, a.ev_stop, 0, k);  \
      else hipLaunchKernelGGL((node_fwd_persistentK_kernel<AA, TT, true>), gridk, blockk, 0, stream, k);                           \
    } else if (a.ev_start) hipExtLaunchKernelGGL((node_fwd_persistentK_kernel<AA, TT>), gridk, blockk, 0, stream, a.ev_start, a.ev_stop, 0, k);  \
    else hipLaunchKernelGGL((node_fwd_persistentK_kernel<AA, TT>), gridk, blockk, 0, stream, k);
    if (a.tape && a.act == NGPDE_ACT_RELU) { NGPDE_PFK_LAUNCH(NGPDE_ACT_RELU, true) }
    else if (a.tape) { NGPDE_PFK_LAUNCH(-1, true) }
    else if (a.act == NGPDE_ACT_RELU) { NGPDE_PFK_LAUNCH(NGPDE_ACT_RELU, false) }
    else { NGPDE_PFK_LAUNCH(-1, false) }
#undef NGPDE_PFK_LAUNCH
    NGPDE_LAUNCH_CHECK("node_fwd_persistentK_kernel");
    if (!a.no_latch) hipLaunchKernelGGL(latch_fault_kernel, dim3(1), dim3(64), 0, stream, k.m.abort_word, ps.fault);
    NGPDE_LAUNCH_CHECK("latch_fault_kernel");
    return turn.leave();
  }
  NGPDE_REQUIRE(!k.m.slot_w || (!a.pair && !a.interleave && a.n_members == 1), NGPDE_ERR_STATE,
                "weighted graphs: one tile per workgroup or tile rounds, one member");
  const dim3 grid(a.pair ? ps.pair_wgs : ps.n_tiles), block(kThreads);
  if (ps.hub) {   // hub geometry: one tile per workgroup and CU
    NGPDE_REQUIRE(!a.pair && !a.interleave, NGPDE_ERR_STATE, "hub geometry: one tile per workgroup, the members of a batch one after the other");
    k.ztape = a.ztape;
    NGPDE_REQUIRE(!a.tape || (a.act == NGPDE_ACT_RELU ? a.masks != nullptr : a.ztape != nullptr), NGPDE_ERR_INVALID_ARGUMENT,
                  "persistent forward with a tape needs the sign-bit masks (relu) or the pre-activation tape");
#define NGPDE_PFH_LAUNCH(AA, TT)                                                                                                   \
    if (a.ev_start) hipExtLaunchKernelGGL((node_fwd_persistent_kernel<AA, TT, false, true>), grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k); \
    else hipLaunchKernelGGL((node_fwd_persistent_kernel<AA, TT, false, true>), grid, block, 0, stream, k);
    if (a.tape && a.act == NGPDE_ACT_RELU) { NGPDE_PFH_LAUNCH(NGPDE_ACT_RELU, true) }
    else if (a.tape) { NGPDE_PFH_LAUNCH(-1, true) }
    else if (a.act == NGPDE_ACT_RELU) { NGPDE_PFH_LAUNCH(NGPDE_ACT_RELU, false) }
    else { NGPDE_PFH_LAUNCH(-1, false) }
#undef NGPDE_PFH_LAUNCH
    NGPDE_LAUNCH_CHECK("node_fwd_persistent_kernel (hub geometry)");
    if (!a.no_latch) hipLaunchKernelGGL(latch_fault_kernel, dim3(1), dim3(64), 0, stream, k.m.abort_word, ps.fault);
    NGPDE_LAUNCH_CHECK("latch_fault_kernel");
    return turn.leave();
  }
#define NGPDE_PF_LAUNCH(AA, TT)                                                                                              \
  if (k.m.slot_w) {                                                                                                          \
    if (a.ev_start) hipExtLaunchKernelGGL((node_fwd_persistent_kernel<AA, TT, true>), grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k); \
    else hipLaunchKernelGGL((node_fwd_persistent_kernel<AA, TT, true>), grid, block, 0, stream, k);                           \
  } else if (a.pair) {                                                                                                              \
    if (a.ev_start) hipExtLaunchKernelGGL((node_fwd_persistent2_kernel<AA, TT, true>), grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k); \
    else hipLaunchKernelGGL((node_fwd_persistent2_kernel<AA, TT, true>), grid, block, 0, stream, k);                          \
  } else if (a.interleave) {                                                                                                 \
    if (a.ev_start) hipExtLaunchKernelGGL((node_fwd_persistent2_kernel<AA, TT, false>), grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k); \
    else hipLaunchKernelGGL((node_fwd_persistent2_kernel<AA, TT, false>), grid, block, 0, stream, k);                         \
  } else if (a.ev_start) hipExtLaunchKernelGGL((node_fwd_persistent_kernel<AA, TT>), grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k); \
  else hipLaunchKernelGGL((node_fwd_persistent_kernel<AA, TT>), grid, block, 0, stream, k);
  k.ztape = a.ztape;
  if (a.tape && a.act == NGPDE_ACT_RELU) {
    NGPDE_REQUIRE(a.masks != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "persistent forward with a relu tape needs the sign-bit masks");
    NGPDE_PF_LAUNCH(NGPDE_ACT_RELU, true)
  } else if (a.tape) {
    NGPDE_REQUIRE(a.ztape != nullptr && !a.interleave && !a.pair, NGPDE_ERR_INVALID_ARGUMENT,
                  "persistent forward with a tape: activations other than relu keep the pre-activations (one member at a time)");
    NGPDE_PF_LAUNCH(-1, true)
  } else if (a.act == NGPDE_ACT_RELU) {
    NGPDE_PF_LAUNCH(NGPDE_ACT_RELU, false)
  } else {
    NGPDE_PF_LAUNCH(-1, false)
  }
#undef NGPDE_PF_LAUNCH
  NGPDE_LAUNCH_CHECK("node_fwd_persistent_kernel");
  if (!a.no_latch) hipLaunchKernelGGL(latch_fault_kernel, dim3(1), dim3(64), 0, stream, k.m.abort_word, ps.fault);
  NGPDE_LAUNCH_CHECK("latch_fault_kernel");
  return turn.leave();
}

int32_t launch_node_bwd_persistent(const NodePersistBwd &a, hipStream_t stream) {
  const ngpde_graph *g = a.g;
  const NodePersist &ps = *a.ps;
  int32_t st;
  PersistentTurn turn;
  if ((st = turn.enter(stream))) return st;
  if ((st = launch_zero(ps.sync, ps.sync_bytes, stream))) return st;
  PBwdK k;
  k.m = make_meta(g->by_s, ps, 1, a.of);
  k.n_steps = a.n_steps; k.S = a.S; k.n_members = a.n_members; k.act = a.act; k.ztape = a.ztape;
  k.lam = a.lam; k.g1 = a.g1; k.g2 = a.g2; k.w1 = a.w1; k.w2 = a.w2;
  k.tape = a.tape; k.masks = a.masks; k.row_elems = a.row_elems; k.mask_bytes = a.mask_bytes;
  k.slab_dw1 = a.slab_dw1; k.slab_db1 = a.slab_db1; k.slab_dw2 = a.slab_dw2; k.slab_db2 = a.slab_db2;
  k.cb = ps.coef + 42;
  k.flag_stride = (size_t)ps.n_tiles * 32;
  k.pair_wgs = a.pair ? ps.pair_wgs : 0;
  k.ubar = a.ubar;
  NGPDE_REQUIRE(!a.pair || (ps.pair_wgs > 0 && !a.interleave && a.n_members == 1 && a.act == NGPDE_ACT_RELU), NGPDE_ERR_STATE,
                "tile-pair launch without its setup");
  k.k_tiles = a.k_tiles;
  if (a.k_tiles > 0) {   // tile rounds
    NGPDE_REQUIRE(ps.pair_wgs > 0 && a.n_members == 1 && a.ubar, NGPDE_ERR_STATE, "tile-round launch without its setup");
    NGPDE_REQUIRE(a.k_tiles <= kMaxTileRounds, NGPDE_ERR_STATE, "tile rounds: at most %d tiles per workgroup", kMaxTileRounds);
    NGPDE_REQUIRE(a.act == NGPDE_ACT_RELU || a.ztape, NGPDE_ERR_INVALID_ARGUMENT, "persistent adjoint: activations other than relu need the saved pre-activations");
    k.pair_wgs = ps.pair_wgs;
    if ((st = launch_zero(a.ubar, 5 * a.row_elems * sizeof(float), stream))) return st;   // the stage adjoints start from zero
    const dim3 gridk(ps.pair_wgs), blockk(kThreads);
    NGPDE_REQUIRE(!k.m.slot_w || a.k_tiles <= kMaxTileRoundsW, NGPDE_ERR_STATE, "weighted tile rounds: at most %d tiles per workgroup", kMaxTileRoundsW);
    const char *nopipe_b = std::getenv("NGPDE_NO_TILE_PIPE");
    const bool piped_b = !k.m.slot_w && !(nopipe_b && nopipe_b[0] == '1');
#define NGPDE_PBK_LAUNCH(AA)                                                                                                      \
    if (piped_b) {                                                                                                                \
      if (a.ev_start) hipExtLaunchKernelGGL((node_bwd_persistentKP_kernel<AA>), gridk, blockk, 0, stream, a.ev_start, a.ev_stop, 0, k);  \
      else hipLaunchKernelGGL((node_bwd_persistentKP_kernel<AA>), gridk, blockk, 0, stream, k);                                    \
    } else if (k.m.slot_w) {                                                                                                      \
      if (a.ev_start) hipExtLaunchKernelGGL((node_bwd_persistentK_kernel<AA, true>), gridk, blockk, 0, stream, a.ev_start, a.ev_stop, 0, k);  \
      else hipLaunchKernelGGL((node_bwd_persistentK_kernel<AA, true>), gridk, blockk, 0, stream, k);                               \
    } else if (a.ev_start) hipExtLaunchKernelGGL((node_bwd_persistentK_kernel<AA, false>), gridk, blockk, 0, stream, a.ev_start, a.ev_stop, 0, k);  \
    else hipLaunchKernelGGL((node_bwd_persistentK_kernel<AA, false>), gridk, blockk, 0, stream, k);
    if (a.act == NGPDE_ACT_RELU) { NGPDE_PBK_LAUNCH(NGPDE_ACT_RELU) }
    else { NGPDE_PBK_LAUNCH(-1) }
#undef NGPDE_PBK_LAUNCH
    NGPDE_LAUNCH_CHECK("node_bwd_persistentK_kernel");
    if (!a.no_latch) hipLaunchKernelGGL(latch_fault_kernel, dim3(1), dim3(64), 0, stream, k.m.abort_word, ps.fault);
    NGPDE_LAUNCH_CHECK("latch_fault_kernel");
    return turn.leave();
  }
  NGPDE_REQUIRE(!k.m.slot_w || (!a.pair && !a.interleave && a.n_members == 1), NGPDE_ERR_STATE,
                "weighted graphs: one tile per workgroup or tile rounds, one member");
  const dim3 grid(a.pair ? ps.pair_wgs : ps.n_tiles), block(kThreads);
  if (ps.hub) {
    NGPDE_REQUIRE(!a.pair && !a.interleave, NGPDE_ERR_STATE, "hub geometry: one tile per workgroup, the members of a batch one after the other");
    NGPDE_REQUIRE(a.act == NGPDE_ACT_RELU || a.ztape, NGPDE_ERR_INVALID_ARGUMENT, "persistent adjoint: activations other than relu need the saved pre-activations");
    if (a.act == NGPDE_ACT_RELU) {
      if (a.ev_start) hipExtLaunchKernelGGL((node_bwd_persistent_kernel<NGPDE_ACT_RELU, false, true>), grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k);
      else hipLaunchKernelGGL((node_bwd_persistent_kernel<NGPDE_ACT_RELU, false, true>), grid, block, 0, stream, k);
    } else {
      if (a.ev_start) hipExtLaunchKernelGGL((node_bwd_persistent_kernel<-1, false, true>), grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k);
      else hipLaunchKernelGGL((node_bwd_persistent_kernel<-1, false, true>), grid, block, 0, stream, k);
    }
  } else if (k.m.slot_w) {
    NGPDE_REQUIRE(a.act == NGPDE_ACT_RELU || a.ztape, NGPDE_ERR_INVALID_ARGUMENT, "persistent adjoint: activations other than relu need the saved pre-activations");
    if (a.act == NGPDE_ACT_RELU) {
      if (a.ev_start) hipExtLaunchKernelGGL((node_bwd_persistent_kernel<NGPDE_ACT_RELU, true>), grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k);
      else hipLaunchKernelGGL((node_bwd_persistent_kernel<NGPDE_ACT_RELU, true>), grid, block, 0, stream, k);
    } else {
      if (a.ev_start) hipExtLaunchKernelGGL((node_bwd_persistent_kernel<-1, true>), grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k);
      else hipLaunchKernelGGL((node_bwd_persistent_kernel<-1, true>), grid, block, 0, stream, k);
    }
  } else if (a.pair) {
    NGPDE_REQUIRE(a.ubar != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "tile-pair persistent adjoint without its stage-adjoint scratch");
    if (a.ev_start) hipExtLaunchKernelGGL(node_bwd_persistent2_kernel<true>, grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k);
    else hipLaunchKernelGGL(node_bwd_persistent2_kernel<true>, grid, block, 0, stream, k);
  } else if (a.interleave) {
    NGPDE_REQUIRE(a.ubar != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "interleaved persistent adjoint without its stage-adjoint scratch");
    if (a.ev_start) hipExtLaunchKernelGGL(node_bwd_persistent2_kernel<false>, grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k);
    else hipLaunchKernelGGL(node_bwd_persistent2_kernel<false>, grid, block, 0, stream, k);
  } else if (a.act != NGPDE_ACT_RELU) {
    NGPDE_REQUIRE(a.ztape != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "persistent adjoint: activations other than relu need the saved pre-activations");
    if (a.ev_start) hipExtLaunchKernelGGL(node_bwd_persistent_kernel<-1>, grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k);
    else hipLaunchKernelGGL(node_bwd_persistent_kernel<-1>, grid, block, 0, stream, k);
  } else if (a.ev_start) hipExtLaunchKernelGGL(node_bwd_persistent_kernel<NGPDE_ACT_RELU>, grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k);
  else hipLaunchKernelGGL(node_bwd_persistent_kernel<NGPDE_ACT_RELU>, grid, block, 0, stream, k);
  NGPDE_LAUNCH_CHECK("node_bwd_persistent_kernel");
  if (!a.no_latch) hipLaunchKernelGGL(latch_fault_kernel, dim3(1), dim3(64), 0, stream, k.m.abort_word, ps.fault);
  NGPDE_LAUNCH_CHECK("latch_fault_kernel");
  return turn.leave();
}

}  // namespace ngpde

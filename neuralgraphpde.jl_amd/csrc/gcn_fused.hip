// gcn_fused.hip -- fused gfx950 kernels of the GCNConv hot path (/root/reference/src/layers.jl:200-239),
// d_in = d_out = D in {16, 32, 64, 128}:
//   forward  : CSR segmented aggregation -> LDS tile -> fp32 MFMA (x W) -> bias/activation
//              [-> Runge-Kutta stage combination], one launch per layer evaluation;
//   backward : [A^T aggregation of the incoming gradient ->] [adjoint stage combination ->] act' mask ->
//              fp32 MFMA (dZ x W^T and X3^T x dZ) -> per-workgroup dW/db slabs.
// No atomics anywhere: every output row / slab has exactly one writer, so results are bitwise
// reproducible run to run (the reference's GPU path scatters with fp32 atomics).
//
// Work decomposition: a workgroup (8 waves) owns one run of kTileRows = 32 schedule entries (a
// graph-compact node cluster, see ngpde_graph::h_order); blockIdx -> run is XCD-aware so each of the 8
// XCDs works on a contiguous range of runs.  N = 16384 gives 512 workgroups = 2 per CU, 4 waves per SIMD.
// The kernels are dependency-latency bound (measured ~1.5k cycles per dependent global round under
// load), so the gather is shaped as TWO rounds: {schedule entry, fixed-width edge entries, own row,
// weights, slab fragments} in the first -- all addressed by schedule position or node-local, none
// depending on another load -- and every neighbour row of a row (up to 16 x 16-byte loads per lane) in
// the second.
#include <hip/hip_ext.h>

#include <cstdlib>

#include "common.h"
#include "device_utils.h"
#include "gcn_tile.h"

namespace ngpde {

namespace {

#ifdef NGPDE_STAMPS
// diagnostic build only (tools/): per-workgroup phase timestamps, one slot of [n_blocks][16] words per
// launch; never compiled into the product library
unsigned long long *g_stamps_base = nullptr;
int g_stamps_max = 0, g_stamps_next = 0;
#define NGPDE_STAMP_FIELD unsigned long long *stamps;
#define NGPDE_STAMP(k)                                                              \
  do {                                                                              \
    if (threadIdx.x == 0 && p.stamps) {                                             \
      p.stamps[(size_t)blockIdx.x * 16 + 2 * (k)] = clock64();                      \
      p.stamps[(size_t)blockIdx.x * 16 + 2 * (k) + 1] = wall_clock64();             \
    }                                                                               \
  } while (0)
#define NGPDE_STAMP_SET(kk, nb)                                                     \
  kk.stamps = nullptr;                                                              \
  if (g_stamps_base && g_stamps_next < g_stamps_max) kk.stamps = g_stamps_base + (size_t)(g_stamps_next++) * (nb) * 16;
#define NGPDE_SUBSTAMP(ptr, k)                                                      \
  do {                                                                              \
    if (threadIdx.x == 0 && (ptr)) (ptr)[(size_t)blockIdx.x * 16 + 10 + (k)] = clock64(); \
  } while (0)
#define NGPDE_USE(v) asm volatile("" ::"v"(v))
#define NGPDE_STAMP_PTR(p) (p).stamps
#else
#define NGPDE_STAMP_FIELD
#define NGPDE_STAMP(k)
#define NGPDE_STAMP_SET(kk, nb)
#define NGPDE_SUBSTAMP(ptr, k)
#define NGPDE_USE(v)
#define NGPDE_STAMP_PTR(p) nullptr
#endif

struct CombDev {
  int n;
  const float *ptr[8];
  float coef[8];
  float coef_self;
};

// v = coef_self * self + sum_k coef[k] * ptr[k][row], split so the (node-local) term loads can be
// requested long before they are combined
__device__ __forceinline__ void comb_prefetch(const CombDev &c, size_t idx4, float4 (&t)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < c.n) t[k] = reinterpret_cast<const float4 *>(c.ptr[k])[idx4];
}
__device__ __forceinline__ float4 comb_finish(const CombDev &c, float4 self, const float4 (&t)[8]) {
  float4 v = f4_scale(c.coef_self, self);
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < c.n) v = f4_fma(c.coef[k], t[k], v);
  return v;
}

// ---- CSR segmented aggregation of whole feature rows ------------------------------------------------
// A "group" of LPR = D/4 adjacent lanes owns R rows; lane q holds features 4q..4q+3 of each row as one
// float4 (a D=64 row = 256 B = 16 lanes x dwordx4: fully coalesced row gathers).  Lane q of the group
// holds entry (base + q) of the row -- {col, coef}, one coalesced 8-byte load -- and the entries are
// broadcast with in-register lane shuffles while R*U independent 16-byte row loads are issued at once.
// Branch-free: slots past the row's degree load row 0 and are zeroed by selects.
//   acc[r] = c * ( sum_e coef_e * X[col_e] + (self ? c * X[node] : 0) ),  c = bits in sched.w
template <int LPR, int R>
__device__ __forceinline__ void load_entries(const int2 *__restrict__ ent, const int4 (&sc)[R], int base, int q,
                                             int (&ecol)[R], int (&ecf)[R]) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool ok = base + q < sc[r].z;
    const int2 v = ent[ok ? sc[r].y + base + q : 0];
    ecol[r] = ok ? v.x : 0;
    ecf[r] = ok ? v.y : 0;
  }
}

template <int LPR, int R, int U>
__device__ __forceinline__ void aggregate_rows(const float4 *__restrict__ X4, const int2 *__restrict__ ent,
                                               int self_loops, const int4 (&sc)[R], int q, int (&ecol)[R],
                                               int (&ecf)[R], const float4 (&selfv)[R], float4 (&acc)[R]) {
  static_assert(LPR % U == 0, "unroll must divide the lanes per row");
  int maxdeg = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    maxdeg = max(maxdeg, sc[r].z);
    acc[r] = f4_zero();
  }
  for (int base = 0; base < maxdeg; base += LPR) {
    if (base > 0) load_entries<LPR, R>(ent, sc, base, q, ecol, ecf);   // chunk 0 arrives preloaded
    const int nin = min(LPR, maxdeg - base);
    for (int e = 0; e < nin; e += U) {
      float4 v[R][U];
      float cf[R][U];
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int col = __shfl(ecol[r], e + u, LPR);         // 0 for slots past the row's degree
          cf[r][u] = __int_as_float(__shfl(ecf[r], e + u, LPR));
          v[r][u] = load_row4<LPR>(X4, col, q);
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool vld = (base + e + u) < sc[r].z;
          float4 t = v[r][u];
          t.x = vld ? t.x : 0.f; t.y = vld ? t.y : 0.f; t.z = vld ? t.z : 0.f; t.w = vld ? t.w : 0.f;
          acc[r] = f4_fma(cf[r][u], t, acc[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float ci = __int_as_float(sc[r].w);
    if (self_loops) acc[r] = f4_fma(ci, selfv[r], acc[r]);
    acc[r] = f4_scale(ci, acc[r]);
  }
}

// Long rows of the per-row gather, cooperatively.  A row's entries are walked by the LPR lanes that own the row, U row fetches in
// flight: a hub of a citation graph (Cora: degree 168 beside a median of 3; BASELINE config 1, graph_node.md:78-83) is 11 dependent
// rounds at D = 64 and 42 at D = 16 while the rest of the workgroup -- and, its tile being the last to finish, the whole launch --
// waits.  Here the rows of a tile with more than kCoopDeg entries (at most kCoopRows of them; more: the tile is dense and the plain
// walk, which serves all rows at once, is the better one) are taken one at a time by kCoopGroups lane groups, each a contiguous run of
// the row's entries; the partial sums meet in LDS and the group that owns the row adds them in group order (fixed: the result does
// not depend on timing; it differs from the serial order in the last bits).  The row leaves with degree 0 and start -1, and coop_add
// puts its sum in after aggregate_rows.  Every thread of the workgroup must call this (barriers); `cnt` is common to the workgroup
// (one int per half), `list` / `part` / `hold` belong to the caller's half.  Measured on the 2 708-node graph of config 1 (Tsit5 x 10
// solve + adjoint on the replayed plan, tools/bench_c1_cora.py): 342 -> 240 us per ODE step at D = 16, 282 -> 240 at D = 32, 297 -> 300
// at D = 64 (seven rounds there; the launches' own floor is ~ 9 us).  Sharing the groups among ALL long rows of a tile in proportion
// to their degrees, in one pass, was built and is slower (277 / 262 / 309: the serial share-out and two more barriers).
constexpr int kCoopDeg = 48, kCoopRows = 4, kCoopGroups = 32;
template <int LPR, int U>
__device__ __forceinline__ float4 gather_run(const float4 *__restrict__ X4, const int2 *__restrict__ ent, int start, int cnt, int q) {
  float4 acc = f4_zero();
  for (int base = 0; base < cnt; base += LPR) {
    const bool ok = base + q < cnt;
    const int2 ev = ent[ok ? start + base + q : 0];
    const int ecol = ok ? ev.x : 0, ecf = ok ? ev.y : 0;
    const int nin = min(LPR, cnt - base);
    for (int e = 0; e < nin; e += U) {
      float4 v[U];
      float cf[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int col = __shfl(ecol, e + u, LPR);
        cf[u] = __int_as_float(__shfl(ecf, e + u, LPR));
        v[u] = load_row4<LPR>(X4, col, q);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool vld = (base + e + u) < cnt;
        float4 t = v[u];
        t.x = vld ? t.x : 0.f; t.y = vld ? t.y : 0.f; t.z = vld ? t.z : 0.f; t.w = vld ? t.w : 0.f;
        acc = f4_fma(cf[u], t, acc);
      }
    }
  }
  return acc;
}

template <int LPR, int R, int U>
__device__ __forceinline__ void coop_long_rows(const float4 *__restrict__ X4, const int2 *__restrict__ ent, int4 (&sc)[R], int grp,
                                               int q, int tid, int half, bool pair, int *cnt, int4 *list, float4 *part,
                                               float4 *hold) {
  constexpr int NGC = (kThreads / LPR < kCoopGroups) ? kThreads / LPR : kCoopGroups;   // D = 128 has 16 groups
  if (tid == 0) cnt[half] = 0;
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (sc[r].x >= 0 && sc[r].z > kCoopDeg) {
        const int at = atomicAdd(&cnt[half], 1);   // (the list's order does not enter the sums)
        if (at < kCoopRows) list[at] = make_int4(grp * R + r, sc[r].y, sc[r].z, 0);
      }
    }
  }
  __syncthreads();
  int mine = cnt[half], other = pair ? cnt[half ^ 1] : 0;
  if (mine > kCoopRows) mine = 0;
  if (other > kCoopRows) other = 0;
  const int n = max(mine, other);   // the same for every thread of the workgroup: both halves pass the same barriers
  for (int i = 0; i < n; ++i) {
    const int4 row = i < mine ? list[i] : make_int4(-1, 0, 0, 0);
    const int per = (row.z + NGC - 1) / NGC;
    if (grp < NGC) {
      const int b = min(grp * per, row.z), e = min(b + per, row.z);
      // (a run is deg / 32 entries: eight fetches in flight are one round up to degree 256; sixteen spill at D = 64)
      part[grp * LPR + q] = gather_run<LPR, (U < 8 ? U : 8)>(X4, ent, row.y + b, e - b, q);
    }
    __syncthreads();
    if (row.x >= 0 && row.x / R == grp) {
      float4 sum = f4_zero();
      for (int g = 0; g < NGC; ++g) sum = f4_add(sum, part[g * LPR + q]);
      hold[row.x * LPR + q] = sum;   // read back by its own thread in coop_add
    }
    __syncthreads();
  }
  if (mine > 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (sc[r].x >= 0 && sc[r].z > kCoopDeg) {
        sc[r].z = 0;    // nothing left for aggregate_rows ...
        sc[r].y = -1;   // ... and coop_add knows the row
      }
    }
  }
}
// acc = c (sum + c self) with the sum formed above: aggregate_rows left c (0 + c self)
template <int LPR, int R>
__device__ __forceinline__ void coop_add(const int4 (&sc)[R], int grp, int q, const float4 *hold, float4 (&acc)[R]) {
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (sc[r].x >= 0 && sc[r].y < 0) acc[r] = f4_fma(__int_as_float(sc[r].w), hold[(grp * R + r) * LPR + q], acc[r]);
}

// First round of a tile's gather for one thread: schedule entries, first entry chunk, own rows.
template <int D>
__device__ __forceinline__ void tile_prologue(const int4 *__restrict__ sched, const int2 *__restrict__ ell,
                                              const int2 *__restrict__ ent, const float4 *__restrict__ X4, int tile,
                                              int grp, int q, bool active, int4 (&sc)[Geo<D>::R],
                                              int (&ecol)[Geo<D>::R], int (&ecf)[Geo<D>::R],
                                              float4 (&selfv)[Geo<D>::R]) {
  using G = Geo<D>;
#pragma unroll
  for (int r = 0; r < G::R; ++r) {
    const size_t pos = (size_t)tile * kTM + grp * G::R + r;
    sc[r] = active ? sched[pos] : make_int4(-1, 0, 0, 0);
    if (G::ELL) {   // position-indexed: does not wait for sc
      const int2 v = active ? ell[pos * kEllWidth + q] : make_int2(0, 0);
      ecol[r] = v.x;
      ecf[r] = v.y;
    }
  }
  if (!G::ELL) load_entries<G::LPR, G::R>(ent, sc, 0, q, ecol, ecf);
#pragma unroll
  for (int r = 0; r < G::R; ++r) selfv[r] = load_row4<G::LPR>(X4, max(sc[r].x, 0), q);
}

// ---- LDS-staged aggregation: HaloRegs / halo_round1 / load_sched / halo_round2 live in gcn_tile.h (shared with gat_fused.hip)
// stage the rows (pre-scaled by c[node]; padding entries have c = 0) and sum every row's neighbours from LDS
// DMA: the rows were sent to LDS by halo_round2 (nothing to write here; the barrier's vmcnt(0) retires them).
// POST: multiply the row sums by c[row] (forward; the pullback of the pre-scaled form has no such factor).
// after_barrier(): loads the caller wants in flight during the LDS pass but NOT in front of the barrier's wait.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int D, bool DMA = false, bool POST = true, class Hook = NoHook, bool SCALE = true>
__device__ __forceinline__ void halo_finish(const HaloRegs<D> &h, bool weighted, int self_loops, int grp, int q,
                                            float *ldsXh, const int4 (&sc)[Geo<D>::R], float4 (&acc)[Geo<D>::R],
                                            unsigned long long *dbg = nullptr, Hook &&after_barrier = NoHook()) {
  using G = Geo<D>;
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
  if constexpr (!DMA) {
    NGPDE_USE(h.hv[0].x); NGPDE_USE(h.hv[G::HI - 1].x);
    NGPDE_SUBSTAMP(dbg, 1);   // halo rows arrived
#pragma unroll
    for (int k = 0; k < G::HI; ++k) {
      const int hh = grp + k * G::GROUPS;
      if (hh < kHaloCap) Xh4[hh * G::LPR + q] = SCALE ? f4_scale(__int_as_float(h.he[k].y), h.hv[k]) : h.hv[k];
    }
  }
  if (grp == 0) Xh4[kHaloCap * G::LPR + q] = f4_zero();
  int wmax = 0;
#pragma unroll
  for (int r = 0; r < G::R; ++r) wmax = max(wmax, sc[r].z);
  // longest row of this wave (uniform): whole 4-slot words beyond it are skipped
  wmax = max(wmax, __shfl_xor(wmax, 16));
  wmax = max(wmax, __shfl_xor(wmax, 32));
  if (G::LPR < 16) {
    wmax = max(wmax, __shfl_xor(wmax, 8));
    wmax = max(wmax, __shfl_xor(wmax, 4));
  }
  wmax = __builtin_amdgcn_readfirstlane(wmax);
  NGPDE_SUBSTAMP(dbg, 2);   // LDS written
  __syncthreads();
  NGPDE_SUBSTAMP(dbg, 3);   // barrier passed
  after_barrier();
#pragma unroll
  for (int r = 0; r < G::R; ++r) {
    const unsigned w[8] = {h.sl[r][0].x, h.sl[r][0].y, h.sl[r][0].z, h.sl[r][0].w,
                           h.sl[r][1].x, h.sl[r][1].y, h.sl[r][1].z, h.sl[r][1].w};
    float4 a = f4_zero();
#pragma unroll
    for (int jw = 0; jw < 8; ++jw) {
      if (jw * 4 < wmax) {   // wave-uniform
        float4 v[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) v[jb] = Xh4[((w[jw] >> (8 * jb)) & 0xff) * G::LPR + q];
        if (weighted) {
          const float4 wv = h.sw[r][jw];
          a = f4_fma(wv.x, v[0], a); a = f4_fma(wv.y, v[1], a); a = f4_fma(wv.z, v[2], a); a = f4_fma(wv.w, v[3], a);
        } else {
          a = f4_add(a, f4_add(f4_add(v[0], v[1]), f4_add(v[2], v[3])));
        }
      }
    }
    const float ci = __int_as_float(sc[r].w);
    if (self_loops) a = f4_add(a, Xh4[min(grp * G::R + r, kTM - 1) * G::LPR + q]);   // own row = slot (position in tile)
    acc[r] = POST ? f4_scale(ci, a) : a;
  }
  NGPDE_USE(acc[0].x);
  NGPDE_SUBSTAMP(dbg, 4);   // LDS aggregation done
}

// ---------------------------------------------------------------------------------------------------
// fused forward:  y = act( (C (A+I) C x) Wt + b ),  optional RK stage combination on the fresh rows
// ---------------------------------------------------------------------------------------------------
struct FwdK {
  const float *x;
  const int4 *sched;
  const int2 *ent, *ell, *halo, *tile_info;
  const uint8_t *slots;
  const float *slot_w;
  int self_loops, n_tiles, act;
  const float *wt, *bias;
  float *y, *save_agg, *save_z;
  int has_comb;
  CombDev comb;
  float *comb_out;
  uint8_t *save_mask;
  int order;   // bit 0: W fetched after the aggregation, bit 1: stage terms fetched after the aggregation.  Both on by default:
               // everything issued at kernel start competes in the memory system with the halo rows the workgroup waits for
               // (measured at C2: layer-1 5.98 -> 5.85 us, layer-2 + stage 7.46 -> 7.06 us)
  NGPDE_STAMP_FIELD
};

// PRE (HALO only): every feature array of the caller's pipeline is stored multiplied by c[row] (x~ = c .* x), so the
// halo rows need no scaling when staged and go to LDS by DMA; y and the stage combination are written in the same form.
template <int D, int ACT, bool HALO, bool PRE>
__global__ __launch_bounds__(kThreads, (D <= 64 ? 4 : 2)) void gcn_fused_fwd_kernel(
    // what the first two rounds of loads need, as leading scalar arguments: they are preloaded into SGPRs at wave launch
    // (-mllvm -amdgpu-kernarg-preload-count, Makefile), so the gather chain does not start behind a kernarg fetch
    const int2 *__restrict__ h_halo, const uint8_t *__restrict__ h_slots, const int4 *__restrict__ h_sched,
    const float *__restrict__ h_x, const float *__restrict__ h_slot_w, int h_n_tiles, const FwdK p) {
  static_assert(!PRE || HALO, "the pre-scaled form exists for the LDS-staged aggregation only");
  using G = Geo<D>;
  // the halo region is dead once the aggregation is done and is re-used for the MFMA output tile
  constexpr int kXZ = (HALO && G::XH > kTM * G::TS) ? G::XH : kTM * G::TS;
  __shared__ __attribute__((aligned(16))) float lds[kXZ + kTM * G::TS + D * G::TS];
  float *ldsXh = lds, *ldsZ = lds, *ldsT = lds + kXZ, *ldsBt = lds + kXZ + kTM * G::TS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = tid / G::LPR, q = tid % G::LPR;
  const int tile = xcd_tile(blockIdx.x, h_n_tiles);
  const int act = ACT >= 0 ? ACT : p.act;
  const bool active = grp * G::R < kTM;
  const float4 *X4 = reinterpret_cast<const float4 *>(h_x);
  NGPDE_STAMP(0);

  // ---- round 1: every load whose address depends on nothing loaded; the gather chain first, pinned in this order
  int4 sc[G::R];
  int ecol[G::R], ecf[G::R];
  float4 selfv[G::R];
  HaloRegs<D> hr;
  if (HALO) {
    halo_round1<D>(h_halo, reinterpret_cast<const uint4 *>(h_slots), reinterpret_cast<const float4 *>(h_slot_w), tile, grp,
                   active, hr);
    load_sched<D>(h_sched, tile, grp, active, sc);
  } else {
    tile_prologue<D>(h_sched, p.ell, p.ent, X4, tile, grp, q, active, sc, ecol, ecf, selfv);
  }
  __builtin_amdgcn_sched_barrier(0);
  // W: B[k = in][j = out] = wt[in][out], stored transposed in LDS: 4 dword loads down a column
  // (coalesced across lanes) -> one ds_write_b128
  float4 wreg[G::NPASS];
  auto load_w = [&]() {
    const int j = tid % D, kg0 = tid / D;
#pragma unroll
    for (int ps = 0; ps < G::NPASS; ++ps) {
      const int kg = kg0 + ps * G::KGP;
      wreg[ps] = f4_zero();
      if (kg < D / 4) {
        const float *w = p.wt + (size_t)(4 * kg) * D + j;
        wreg[ps] = make_float4(w[0], w[D], w[2 * D], w[3 * D]);
      }
    }
  };
  if (!(p.order & 1)) load_w();
  const float4 b4 = p.bias ? reinterpret_cast<const float4 *>(p.bias)[q] : f4_zero();   // uniform condition
  __builtin_amdgcn_sched_barrier(0);
  // ---- round 2: addresses from round 1 -- the tile's distinct rows, then the node-local stage terms
  if (HALO) {
    NGPDE_USE(hr.he[0].x);
    NGPDE_SUBSTAMP(NGPDE_STAMP_PTR(p), 0);   // round-1 data arrived
    halo_round2<D, PRE>(X4, q, grp, ldsXh, hr);
  }
  float4 cterm[G::R][8];
  if (HALO && active && p.has_comb && !(p.order & 2)) {
#pragma unroll
    for (int r = 0; r < G::R; ++r) comb_prefetch(p.comb, (size_t)max(sc[r].x, 0) * G::LPR + q, cterm[r]);
  }
  __builtin_amdgcn_sched_barrier(0);
  float4 acc[G::R];
  if (HALO) {
    halo_finish<D, PRE, true>(hr, h_slot_w != nullptr, p.self_loops, grp, q, ldsXh, sc, acc, NGPDE_STAMP_PTR(p));
  } else {
    // (LDS: the product's operand tiles are not written yet -- W waits in registers, the row sums are what this forms)
    // counter + row list (20 words) in W's tile, partial sums in the row tile, the rows' sums in the result tile (kTM x D floats each)
    coop_long_rows<G::LPR, G::R, G::U>(X4, p.ent, sc, grp, q, tid, 0, false, reinterpret_cast<int *>(ldsBt),
                                       reinterpret_cast<int4 *>(ldsBt + 4), reinterpret_cast<float4 *>(ldsT),
                                       reinterpret_cast<float4 *>(ldsZ));
    aggregate_rows<G::LPR, G::R, G::U>(X4, p.ent, p.self_loops, sc, q, ecol, ecf, selfv, acc);
    coop_add<G::LPR, G::R>(sc, grp, q, reinterpret_cast<const float4 *>(ldsZ), acc);
  }
  NGPDE_STAMP(1);
  if (p.order & 1) load_w();
  if (HALO && active && p.has_comb && (p.order & 2)) {
#pragma unroll
    for (int r = 0; r < G::R; ++r) comb_prefetch(p.comb, (size_t)max(sc[r].x, 0) * G::LPR + q, cterm[r]);
  }
  if (!HALO && active && p.has_comb) {   // the per-row gather keeps 16 rows in flight: stage terms only afterwards
#pragma unroll
    for (int r = 0; r < G::R; ++r) comb_prefetch(p.comb, (size_t)max(sc[r].x, 0) * G::LPR + q, cterm[r]);
  }
  if (active) {
#pragma unroll
    for (int r = 0; r < G::R; ++r) {
      if (sc[r].x < 0) acc[r] = f4_zero();
      *reinterpret_cast<float4 *>(&ldsT[(grp * G::R + r) * G::TS + 4 * q]) = acc[r];
      if (p.save_agg && sc[r].x >= 0) store_stream4(&reinterpret_cast<float4 *>(p.save_agg)[(size_t)sc[r].x * G::LPR + q], acc[r]);
    }
  }
  {
    const int j = tid % D, kg0 = tid / D;
#pragma unroll
    for (int ps = 0; ps < G::NPASS; ++ps) {
      const int kg = kg0 + ps * G::KGP;
      if (kg < D / 4) *reinterpret_cast<float4 *>(&ldsBt[j * G::TS + 4 * kg]) = wreg[ps];
    }
  }
  __syncthreads();
  NGPDE_STAMP(2);
  mfma_rows_times_bt<D>(ldsT, ldsBt, ldsZ, wave_u, lane);
  __syncthreads();
  NGPDE_STAMP(3);
  if (active) {
#pragma unroll
    for (int r = 0; r < G::R; ++r) {
      if (sc[r].x < 0) continue;
      const size_t idx4 = (size_t)sc[r].x * G::LPR + q;
      const float4 z = f4_add(*reinterpret_cast<const float4 *>(&ldsZ[(grp * G::R + r) * G::TS + 4 * q]), b4);
      if (p.save_z) reinterpret_cast<float4 *>(p.save_z)[idx4] = z;
      if (p.save_mask)   // relu'(z) for the pullback: 4 bits instead of re-reading 16 bytes of y
        p.save_mask[((size_t)tile * G::R + r) * kThreads + tid] =
            (uint8_t)((z.x > 0.f ? 1 : 0) | (z.y > 0.f ? 2 : 0) | (z.z > 0.f ? 4 : 0) | (z.w > 0.f ? 8 : 0));
      float4 yv = f4_act(act, z);
      if (PRE) yv = f4_scale(__int_as_float(sc[r].w), yv);
      reinterpret_cast<float4 *>(p.y)[idx4] = yv;
      // (written non-temporally the stage input leaves this launch 0.35 us earlier and reaches the next launch's gather
      // 0.3 us later: a wash, unlike g_out in the pullback)
      if (p.has_comb) reinterpret_cast<float4 *>(p.comb_out)[idx4] = comb_finish(p.comb, yv, cterm[r]);
    }
  }
  NGPDE_STAMP(4);
}

// ---------------------------------------------------------------------------------------------------
// fused backward of one layer evaluation
// ---------------------------------------------------------------------------------------------------
struct BwdK {
  const float *g_in;
  const int4 *sched;
  const int2 *ent, *ell, *halo, *tile_info;
  const uint8_t *slots;
  const float *slot_w;
  int self_loops, n_tiles, act;
  int has_comb;
  CombDev comb;
  float *store_t, *store_v;
  float v_scale;
  int do_dense, tape_late, order;
  const float *z, *saved_agg, *wt;
  const uint8_t *mask;
  float *g_out, *slab_dw, *slab_db;
  NGPDE_STAMP_FIELD
};

// Slab layout (per workgroup): dW as [tile tt = mt * CT + nt][lane][4] (each lane's four MFMA result
// registers contiguous -> one 16-byte read-modify-write per tile), db as [D].
//
// PAIR: a 1024-thread workgroup = two independent 512-thread halves, each with its own tile and LDS region, that fold their
// dW / db partials into ONE slab (half 1 hands its accumulators to half 0 through LDS).  Halves the slab read-modify-write
// traffic of every backward launch (16.8 MB of the 36-55 MB a launch moves at C2) at the price of one more barrier.
//
// PRE: the pre-scaled form of the caller's pipeline (see the forward kernel): g_in holds c .* G, so the halo rows go to LDS
// by DMA and the row sums carry no c factor; c[row] enters where dL/dy = c .* dL/dy~ (before the mask) and where the
// product G is stored for the next gather.
template <int D, bool AGG, int ACT, bool HALO, bool PAIR, bool PRE>
__global__ __launch_bounds__((PAIR ? 2 : 1) * kThreads, (PAIR || D <= 64 ? 4 : 2)) void gcn_fused_bwd_kernel(
    const int2 *__restrict__ h_halo, const uint8_t *__restrict__ h_slots, const int4 *__restrict__ h_sched,
    const float *__restrict__ h_x, const float *__restrict__ h_slot_w, int h_n_tiles, const BwdK p) {
  using G = Geo<D>;
  static_assert(!PRE || !AGG || HALO, "the pre-scaled form exists for the LDS-staged aggregation only");
#ifdef NGPDE_PRE_NO_BWD_DMA
  constexpr bool DMA = false;
#else
  constexpr bool DMA = PRE && AGG && HALO;
#endif
  constexpr int kXZ = (AGG && HALO && G::XH > kTM * G::TS) ? G::XH : kTM * G::TS;
  constexpr int kRegion = kXZ + kTM * G::TS * 2 + D * G::TS;
  __shared__ __attribute__((aligned(16))) float lds_all[(PAIR ? 2 : 1) * kRegion];
  const int half = PAIR ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 9)) : 0;
  // the halo regions of both halves first: an LDS-DMA destination is a 16-bit offset (M0[15:0]), it must stay below 64 KiB
  constexpr int kRest = kRegion - kXZ;
  float *ldsXh = lds_all + half * kXZ, *ldsG = ldsXh;
  float *rest = lds_all + (PAIR ? 2 : 1) * kXZ + half * kRest;
  float *ldsDZ = rest, *ldsX = rest + kTM * G::TS, *ldsBt = rest + 2 * kTM * G::TS;
  static_assert(!DMA || (PAIR ? 2 : 1) * kXZ * sizeof(float) <= 65536, "LDS-DMA destinations must lie in the first 64 KiB");
  const int tid = PAIR ? (threadIdx.x & (kThreads - 1)) : threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = tid / G::LPR, q = tid % G::LPR;
  // PAIR: workgroup b owns tiles 2 b' and 2 b' + 1 (b' XCD-mapped); the second half of an odd last pair repeats the last
  // tile with every row masked out (it contributes zeros and stores nothing)
  const int n_wg = PAIR ? (h_n_tiles + 1) / 2 : h_n_tiles;
  const int tile_raw = PAIR ? 2 * xcd_tile(blockIdx.x, n_wg) + half : xcd_tile(blockIdx.x, n_wg);
  const bool tile_ok = tile_raw < h_n_tiles;
  const int tile = tile_ok ? tile_raw : h_n_tiles - 1;
  const int act = ACT >= 0 ? ACT : p.act;
  const bool active = grp * G::R < kTM;
  const float4 *G4 = reinterpret_cast<const float4 *>(h_x);
  NGPDE_STAMP(0);

  // ---- round 1 / round 2 of the gather chain first (vmcnt retires in order: slower loads must not sit in front)
  int4 sc[G::R];
  int ecol[G::R], ecf[G::R];
  float4 selfv[G::R];
  HaloRegs<D> hr;
  if (AGG && HALO) {
    halo_round1<D>(h_halo, reinterpret_cast<const uint4 *>(h_slots), reinterpret_cast<const float4 *>(h_slot_w), tile, grp,
                   active, hr);
    load_sched<D>(h_sched, tile, grp, active, sc);
    halo_round2<D, DMA>(G4, q, grp, ldsXh, hr);
  } else if (AGG) {
    tile_prologue<D>(h_sched, p.ell, p.ent, G4, tile, grp, q, active, sc, ecol, ecf, selfv);
  } else {
    load_sched<D>(h_sched, tile, grp, active, sc);
  }
  if (PAIR && !tile_ok) {
#pragma unroll
    for (int r = 0; r < G::R; ++r) sc[r].x = -1;
  }
  // relu sign bits written by the forward launch of the same layer evaluation: addressed by (tile, thread), no dependency
  unsigned mk[G::R];
  auto load_mask = [&]() {
#pragma unroll
    for (int r = 0; r < G::R; ++r) mk[r] = (p.mask && p.do_dense) ? p.mask[((size_t)tile * G::R + r) * kThreads + tid] : 0u;
  };
  // DMA: the barrier that publishes the staged rows waits for EVERY outstanding load of the wave (vmcnt(0)), so the
  // node-local loads that register staging issues up front go out right behind that barrier instead and fly during
  // the LDS pass.
  if (!DMA) load_mask();
  // B = Wt^T : B[k = o][j = i] = wt[i][o]  ->  Bt[j = i][k = o] = wt[i][o]: a straight copy
  float4 wreg[G::W4];
  // slab fragments of this wave's dW tiles and this thread's db column: consumed after the MFMAs
  constexpr int NT = G::CT * G::CT;
  float4 *slab4 = reinterpret_cast<float4 *>(p.slab_dw + (size_t)blockIdx.x * D * D);
  float4 sl[G::DWT];
  const int dbc = tid / G::DBP, dbpart = tid % G::DBP;
  float dbv = 0.f;
  auto load_w = [&]() {
    if (p.do_dense) {
#pragma unroll
      for (int k = 0; k < G::W4; ++k) {
        const int idx = tid + k * kThreads;
        wreg[k] = (idx < D * D / 4) ? reinterpret_cast<const float4 *>(p.wt)[idx] : f4_zero();
      }
    }
  };
  if (!DMA && !(p.order & 1)) load_w();
  auto load_slab = [&]() {   // consumed only after the dW MFMAs
    if (p.do_dense) {
#pragma unroll
      for (int m = 0; m < G::DWT; ++m) {
        const int tt = wave_u + G::WAVES * m;
        sl[m] = (tt < NT && half == 0) ? slab4[tt * 64 + lane] : f4_zero();
      }
      if (dbpart == 0 && half == 0) dbv = p.slab_db[(size_t)blockIdx.x * D + dbc];
    }
  };
  if (!DMA && !(p.order & 2)) load_slab();
  float4 zrow[G::R], xrow[G::R], cterm[G::R][8];
  auto load_tape = [&]() {   // saved activations of this thread's rows (node-local, HBM-resident tape)
    if (active && p.do_dense) {
#pragma unroll
      for (int r = 0; r < G::R; ++r) {
        const size_t idx4 = (size_t)max(sc[r].x, 0) * G::LPR + q;
        if (!p.mask) zrow[r] = load_stream4(&reinterpret_cast<const float4 *>(p.z)[idx4]);
        xrow[r] = load_stream4(&reinterpret_cast<const float4 *>(p.saved_agg)[idx4]);
      }
    }
  };
  // Without stage terms the tape rows are fetched only after the aggregation: issued up front they queue in the memory
  // system in front of the halo rows the whole workgroup waits for (measured: -0.7 us per launch); with stage terms
  // (a later batch of node-local loads anyway) fetching them early is the faster order.
  if (!DMA && !p.tape_late) load_tape();

  float4 t[G::R];
  if (AGG && HALO) {
    if constexpr (DMA) {
      halo_finish<D, true, false>(hr, h_slot_w != nullptr, p.self_loops, grp, q, ldsXh, sc, t, nullptr, [&]() {
        load_mask();
        if (!(p.order & 2)) load_slab();
        load_tape();
      });
    } else {
      halo_finish<D, false, !PRE, NoHook, !PRE>(hr, h_slot_w != nullptr, p.self_loops, grp, q, ldsXh, sc, t);
    }
  } else if (AGG) {
    // the counters (one per half, read by both) in the first half's W tile; per half: row list in the X tile, partial sums in the dz
    // tile, the rows' sums in the result tile
    coop_long_rows<G::LPR, G::R, G::U>(G4, p.ent, sc, grp, q, tid, half, PAIR,
                                       reinterpret_cast<int *>(lds_all + (PAIR ? 2 : 1) * kXZ + 2 * kTM * G::TS),
                                       reinterpret_cast<int4 *>(ldsX), reinterpret_cast<float4 *>(ldsDZ),
                                       reinterpret_cast<float4 *>(ldsXh));
    aggregate_rows<G::LPR, G::R, G::U>(G4, p.ent, p.self_loops, sc, q, ecol, ecf, selfv, t);
    coop_add<G::LPR, G::R>(sc, grp, q, reinterpret_cast<const float4 *>(ldsXh), t);
  } else {
#pragma unroll
    for (int r = 0; r < G::R; ++r) t[r] = G4[(size_t)max(sc[r].x, 0) * G::LPR + q];
  }
  NGPDE_STAMP(1);
  if (!DMA && p.tape_late) load_tape();
  if (DMA || (p.order & 1)) load_w();
  if (p.order & 2) load_slab();
  if (active && p.has_comb) {   // adjoint stage terms: one batch of independent node-local loads
#pragma unroll
    for (int r = 0; r < G::R; ++r) comb_prefetch(p.comb, (size_t)max(sc[r].x, 0) * G::LPR + q, cterm[r]);
  }
  if (active) {
#pragma unroll
    for (int r = 0; r < G::R; ++r) {
      const bool ok = sc[r].x >= 0;
      const size_t idx4 = (size_t)max(sc[r].x, 0) * G::LPR + q;
      float4 kbar = t[r];
      if (ok) {
        if (p.store_t) reinterpret_cast<float4 *>(p.store_t)[idx4] = t[r];
        if (p.has_comb) {
          const float4 v = comb_finish(p.comb, t[r], cterm[r]);
          if (p.store_v) reinterpret_cast<float4 *>(p.store_v)[idx4] = v;
          kbar = f4_scale(p.v_scale, v);
        }
      }
      if (p.do_dense) {
        float4 dz = f4_zero(), xa = f4_zero();
        if (PRE) kbar = f4_scale(__int_as_float(sc[r].w), kbar);   // dL/dy = c .* dL/dy~
        if (ok) {
          if (p.mask)
            dz = make_float4((mk[r] & 1u) ? kbar.x : 0.f, (mk[r] & 2u) ? kbar.y : 0.f, (mk[r] & 4u) ? kbar.z : 0.f,
                             (mk[r] & 8u) ? kbar.w : 0.f);
          else
            dz = f4_mul(kbar, f4_dact(act, zrow[r]));
          xa = xrow[r];
        }
        *reinterpret_cast<float4 *>(&ldsDZ[(grp * G::R + r) * G::TS + 4 * q]) = dz;
        *reinterpret_cast<float4 *>(&ldsX[(grp * G::R + r) * G::TS + 4 * q]) = xa;
      }
    }
  }
  if (!p.do_dense) return;  // uniform for the whole grid
#pragma unroll
  for (int k = 0; k < G::W4; ++k) {
    const int idx = tid + k * kThreads;
    if (idx < D * D / 4) {
      const int wi = (idx * 4) / D, wo = (idx * 4) % D;
      *reinterpret_cast<float4 *>(&ldsBt[wi * G::TS + wo]) = wreg[k];
    }
  }
  __syncthreads();
  NGPDE_STAMP(2);
  // G = dZ x Wt^T  (gradient w.r.t. the aggregated input)
  mfma_rows_times_bt<D>(ldsDZ, ldsBt, ldsG, wave_u, lane);
  __syncthreads();   // G complete in LDS; W^T no longer needed
  NGPDE_STAMP(3);
  // the product rows leave for memory now and drain under the dW product instead of at the end of the launch
  if (active) {
#pragma unroll
    for (int r = 0; r < G::R; ++r) {
      if (sc[r].x < 0) continue;
      float4 gv = *reinterpret_cast<const float4 *>(&ldsG[(grp * G::R + r) * G::TS + 4 * q]);
      if (PRE) gv = f4_scale(__int_as_float(sc[r].w), gv);   // stored as c .* G: the next launch gathers it raw
      store_stream4(&reinterpret_cast<float4 *>(p.g_out)[(size_t)sc[r].x * G::LPR + q], gv);   // gathered once by the next launch
    }
  }
  NGPDE_STAMP(4);
  // dWt[i][o] += sum_n X3[n][i] dZ[n][o]   (K = kTM rows of this tile)
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int m = 0; m < G::DWT; ++m) {
    const int tt = wave_u + G::WAVES * m;
    if (tt < NT) {   // wave-uniform
      const int mt = tt / G::CT, nt = tt % G::CT;
      float a[kTM / 4], b[kTM / 4];
#pragma unroll
      for (int ks = 0; ks < kTM / 4; ++ks) {
        a[ks] = ldsX[(4 * ks + kq) * G::TS + mt * 16 + i];
        b[ks] = ldsDZ[(4 * ks + kq) * G::TS + nt * 16 + i];
      }
      f32x4 acc = (f32x4){sl[m].x, sl[m].y, sl[m].z, sl[m].w};
#pragma unroll
      for (int ks = 0; ks < kTM / 4; ++ks) acc = mfma16(a[ks], b[ks], acc);
      sl[m] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      if (!PAIR) slab4[tt * 64 + lane] = sl[m];
    }
  }
  // db += column sums of the dZ tile: DBP adjacent lanes hold row-partials of one column
  {
    float s = 0.f;
#pragma unroll
    for (int n = dbpart; n < kTM; n += G::DBP) s += ldsDZ[n * G::TS + dbc];
#pragma unroll
    for (int o = 1; o < G::DBP; o <<= 1) s += __shfl_xor(s, o);
    dbv += s;
    if (!PAIR && dbpart == 0) p.slab_db[(size_t)blockIdx.x * D + dbc] = dbv;
  }
  NGPDE_STAMP(5);
  if (PAIR) {
    // half 1 parks its dW tiles and db partials in its W^T region (idle since the barrier above), half 0 folds them in
    float *park = lds_all + 2 * kXZ + kRest + 2 * kTM * G::TS;   // [NT][64 lanes][4] + [D]
    if (half == 1) {
#pragma unroll
      for (int m = 0; m < G::DWT; ++m) {
        const int tt = wave_u + G::WAVES * m;
        if (tt < NT) reinterpret_cast<float4 *>(park)[tt * 64 + lane] = sl[m];
      }
      if (dbpart == 0) park[NT * 256 + dbc] = dbv;
    }
    __syncthreads();
    if (half == 0) {   // fixed order: (slab + half 0) + half 1
#pragma unroll
      for (int m = 0; m < G::DWT; ++m) {
        const int tt = wave_u + G::WAVES * m;
        if (tt < NT) slab4[tt * 64 + lane] = f4_add(sl[m], reinterpret_cast<const float4 *>(park)[tt * 64 + lane]);
      }
      if (dbpart == 0) p.slab_db[(size_t)blockIdx.x * D + dbc] = dbv + park[NT * 256 + dbc];
    }
  }
  NGPDE_STAMP(6);
}

// ---------------------------------------------------------------------------------------------------
// GAT-style softmax aggregation on the same tile / halo machinery (heads * c == 64, heads in {1, 2, 4}):
//   out_i = sum_{e: t_e = i} alpha_e Wx[s_e],   alpha = softmax over the row of leakyrelu(al_i + ar_{s_e}) per head
// The tile's distinct source rows of Wx and their ar scores are staged in LDS once; lane group g owns row g, lane q the
// features 4q..4q+3 (all of one head); max / denominator / weighted sum are three short LDS passes over the row's slots.
// alpha is written once (p order) for the pullback.  [GraphNeuralNetworks.jl GATConv; BASELINE config 3]
// ---------------------------------------------------------------------------------------------------
struct GatK {
  const float *wx, *al, *ar;
  const int4 *sched;
  const int2 *halo;
  const uint8_t *slots;
  int n_tiles, heads;
  float slope;
  float *out, *alpha;
};

__global__ __launch_bounds__(kThreads, 4) void gat_fused_fwd_kernel(const GatK p) {
  constexpr int D = 64;
  using G = Geo<D>;
  __shared__ __attribute__((aligned(16))) float ldsXh[(kHaloCap + 1) * D];
  __shared__ __attribute__((aligned(16))) float ldsAr[(kHaloCap + 1) * 4];
  const int tid = threadIdx.x;
  const int grp = tid / G::LPR, q = tid % G::LPR;
  const int tile = xcd_tile(blockIdx.x, p.n_tiles);
  const bool active = grp * G::R < kTM;
  const float4 *X4 = reinterpret_cast<const float4 *>(p.wx);
  const int H = p.heads, kh = (4 * q) / (D / H);      // this lane's head

  HaloRegs<D> hr;
  int4 sc[G::R];
  halo_round1<D>(p.halo, reinterpret_cast<const uint4 *>(p.slots), nullptr, tile, grp, active, hr);
  load_sched<D>(p.sched, tile, grp, active, sc);
  halo_round2<D, true>(X4, q, grp, ldsXh, hr);   // Wx rows are staged as they are: straight to LDS
  // ar of the halo nodes (lanes q < H fetch one score each), al of this row's head
  float arv[G::HI];
#pragma unroll
  for (int k = 0; k < G::HI; ++k) arv[k] = (q < H) ? p.ar[(size_t)hr.he[k].x * H + q] : 0.f;
  float ali[G::R];
#pragma unroll
  for (int r = 0; r < G::R; ++r) ali[r] = p.al[(size_t)max(sc[r].x, 0) * H + kh];

  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
#pragma unroll
  for (int k = 0; k < G::HI; ++k) {
    const int hh = grp + k * G::GROUPS;
    if (hh < kHaloCap && q < 4) ldsAr[hh * 4 + q] = arv[k];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < G::R; ++r) {
    if (!active || sc[r].x < 0) continue;
    const unsigned w[8] = {hr.sl[r][0].x, hr.sl[r][0].y, hr.sl[r][0].z, hr.sl[r][0].w,
                           hr.sl[r][1].x, hr.sl[r][1].y, hr.sl[r][1].z, hr.sl[r][1].w};
    const int deg = sc[r].z;
    auto slot_of = [&](int j) {
      unsigned word = 0;
#pragma unroll
      for (int jw = 0; jw < 8; ++jw) word = (jw == (j >> 2)) ? w[jw] : word;
      return (int)((word >> (8 * (j & 3))) & 0xff);
    };
    auto score = [&](int sl) {
      const float v = ali[r] + ldsAr[sl * 4 + kh];
      return v > 0.f ? v : p.slope * v;
    };
    float mx = -INFINITY;
    for (int j = 0; j < deg; ++j) mx = fmaxf(mx, score(slot_of(j)));
    float sum = 0.f;
    float4 acc = f4_zero();
    for (int j = 0; j < deg; ++j) {
      const int sl = slot_of(j);
      const float e = fast_exp(score(sl) - mx);
      sum += e;
      acc = f4_fma(e, Xh4[sl * G::LPR + q], acc);
    }
    const float inv = deg > 0 ? fast_rcp(sum) : 0.f;
    reinterpret_cast<float4 *>(p.out)[(size_t)sc[r].x * G::LPR + q] = f4_scale(inv, acc);
    if ((4 * q) % (D / H) == 0)                      // one lane per head writes the row's coefficients
      for (int j = 0; j < deg; ++j) p.alpha[(size_t)(sc[r].y + j) * H + kh] = fast_exp(score(slot_of(j)) - mx) * inv;
  }
}

// out (row-major dWt[i][o], or db[o] when ct == 0) = sum over slabs: 64 elements x 16 slab lanes per workgroup, four
// independent partial sums per lane, combined in a fixed order
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float *__restrict__ slab, int n_slabs, int len, int ct,
                                                            float *__restrict__ out) {
  __shared__ float part[16][64];
  const int el = threadIdx.x & 63, part_id = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < len) {
    int b = part_id;
    for (; b + 48 < n_slabs; b += 64) {
      s0 += slab[(size_t)b * len + e];
      s1 += slab[(size_t)(b + 16) * len + e];
      s2 += slab[(size_t)(b + 32) * len + e];
      s3 += slab[(size_t)(b + 48) * len + e];
    }
    for (; b < n_slabs; b += 16) s0 += slab[(size_t)b * len + e];
  }
  part[part_id][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part_id == 0 && e < len) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[k][el];
    int dst = e;
    if (ct > 0) {  // e = (tt * 64 + lane) * 4 + reg  ->  dWt[(mt*16 + 4*kq + reg) * D + nt*16 + i]
      const int reg = e & 3, ln = (e >> 2) & 63, tt = e >> 8;
      const int mt = tt / ct, nt = tt % ct;
      dst = (mt * 16 + 4 * (ln >> 4) + reg) * (ct * 16) + nt * 16 + (ln & 15);
    }
    out[dst] = v;
  }
}

CombDev to_dev(const Comb &c) {
  CombDev d;
  d.n = c.n;
  for (int k = 0; k < 8; ++k) {
    d.ptr[k] = c.ptr[k];
    d.coef[k] = c.coef[k];
  }
  d.coef_self = c.coef_self;
  return d;
}

// NGPDE_NO_HALO=1 forces the per-row global gather (A/B measurements and tests of the fallback path)
inline bool no_halo_env() {
  static const bool v = [] { const char *e = std::getenv("NGPDE_NO_HALO"); return e && e[0] == '1'; }();
  return v;
}

// paired workgroups for the backward kernels (D <= 64: two 60 KB regions fit the CU's LDS); NGPDE_NO_PAIR=1 for A/B runs
inline bool fused_bwd_pairs(int d) {
  static const bool off = [] { const char *e = std::getenv("NGPDE_NO_PAIR"); return e && e[0] == '1'; }();
  return d <= 64 && !off;
}

// activations with a compiled-in fast path; everything else takes the runtime switch (ACT = -1)
inline int act_template(int act) { return (act == NGPDE_ACT_RELU || act == NGPDE_ACT_IDENTITY) ? act : -1; }

}  // namespace

#ifdef NGPDE_STAMPS
extern "C" int32_t ngpde_debug_set_stamps(unsigned long long *dev_buf, int32_t max_launches) {
  g_stamps_base = dev_buf;
  g_stamps_max = max_launches;
  g_stamps_next = 0;
  return NGPDE_OK;
}
#endif

// the pre-scaled pipeline (rows stored as c .* x, gathered raw by LDS-DMA): every tile staged from LDS in both directions,
// and c finite and positive (self loops: degree >= 1)
bool fused_prescaled_supported(const ngpde_graph *g, int d) {
  return g && g->has_norm && g->self_loops && d <= 128 && fused_supported(d, d) && g->by_t.halo_ok && g->by_s.halo_ok &&
         !no_halo_env();
}
bool fused_supported(int din, int dout) { return din == dout && (din == 16 || din == 32 || din == 64 || din == 128); }
int fused_tile_rows() { return kTM; }
int fused_num_blocks(int64_t n_nodes) { return (int)((n_nodes + kTM - 1) / kTM); }
size_t fused_mask_bytes(int64_t n_nodes, int d) {
  const int lpr = d / 4, groups = kThreads / lpr, r = (kTM + groups - 1) / groups;   // Geo<d>::R
  return (size_t)fused_num_blocks(n_nodes) * r * kThreads;
}
// slabs the backward launches write (paired workgroups share one)
int fused_num_slabs(int64_t n_nodes, int d) {
  const int nb = fused_num_blocks(n_nodes);
  return fused_bwd_pairs(d) ? (nb + 1) / 2 : nb;
}

int32_t launch_fused_fwd(const FusedFwdArgs &a, hipStream_t stream) {
  const ngpde_graph *g = a.g;
  NGPDE_REQUIRE(g && g->has_norm, NGPDE_ERR_STATE, "GCN normalisation not set (call ngpde_graph_set_gcn_norm)");
  NGPDE_REQUIRE(fused_supported(a.d, a.d), NGPDE_ERR_UNSUPPORTED, "fused GCN path needs d in {16,32,64,128}, got %d", a.d);
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE((uint64_t)g->n_nodes * a.d * 4 < (1ull << 32), NGPDE_ERR_UNSUPPORTED,
                "fused GCN path addresses feature arrays with 32-bit byte offsets: n_nodes * d * 4 must be < 4 GiB");
  FwdK k;
  k.x = a.x; k.sched = g->by_t.sched; k.ent = g->by_t.ent; k.ell = g->by_t.ell;
  k.halo = g->by_t.halo; k.tile_info = g->by_t.tile_info; k.slots = g->by_t.slots; k.slot_w = g->by_t.slot_w;
  if (a.of && a.of->slots[0] && g->by_t.halo_ok && !no_halo_env()) {   // a solver plan's own-first tables (LDS-staged aggregation only:
    k.slots = a.of->slots[0]; k.sched = a.of->sched[0];                 // the same rows in another order, padded lengths in the schedule)
    if (k.slot_w) k.slot_w = a.of->slot_w[0];
  }
  k.self_loops = g->self_loops; k.n_tiles = fused_num_blocks(g->n_nodes); k.act = a.act;
  k.wt = a.wt; k.bias = a.bias; k.y = a.y; k.save_agg = a.save_agg; k.save_z = a.save_z;
  k.has_comb = a.has_comb ? 1 : 0; k.comb = to_dev(a.comb); k.comb_out = a.comb_out;
  k.save_mask = a.save_mask;
  {   // default 3; NGPDE_FWD_ORDER=0..3 for A/B runs of the load placement
    static const int ord = [] { const char *e = std::getenv("NGPDE_FWD_ORDER"); return e ? atoi(e) : 3; }();
    k.order = ord;
  }
  NGPDE_STAMP_SET(k, k.n_tiles)
  const bool use_halo = g->by_t.halo_ok && !no_halo_env();
  NGPDE_REQUIRE(!a.pre || fused_prescaled_supported(g, a.d), NGPDE_ERR_UNSUPPORTED,
                "the pre-scaled form needs self loops and tiles that fit the LDS halo in both directions");
  const dim3 grid(k.n_tiles), block(kThreads);
#define NGPDE_FWD_LAUNCH2(DD, AA, HH, PP)                                                                          \
  if (a.ev_start) hipExtLaunchKernelGGL((gcn_fused_fwd_kernel<DD, AA, HH, PP>), grid, block, 0, stream, a.ev_start, a.ev_stop, 0, \
                                        k.halo, k.slots, k.sched, k.x, k.slot_w, k.n_tiles, k);                       \
  else hipLaunchKernelGGL((gcn_fused_fwd_kernel<DD, AA, HH, PP>), grid, block, 0, stream, k.halo, k.slots, k.sched, k.x, k.slot_w, k.n_tiles, k);
#define NGPDE_FWD_LAUNCH(DD, AA)                                                                                   \
  if (Geo<DD>::HALO && a.pre) { NGPDE_FWD_LAUNCH2(DD, AA, (Geo<DD>::HALO), (Geo<DD>::HALO)) }                      \
  else if (Geo<DD>::HALO && use_halo) { NGPDE_FWD_LAUNCH2(DD, AA, (Geo<DD>::HALO), false) }                        \
  else { NGPDE_FWD_LAUNCH2(DD, AA, false, false) }
#define NGPDE_FWD_CASE(DD)                                                            \
  case DD:                                                                            \
    switch (act_template(a.act)) {                                                    \
      case NGPDE_ACT_RELU: NGPDE_FWD_LAUNCH(DD, NGPDE_ACT_RELU) break;                \
      case NGPDE_ACT_IDENTITY: NGPDE_FWD_LAUNCH(DD, NGPDE_ACT_IDENTITY) break;        \
      default: NGPDE_FWD_LAUNCH(DD, -1) break;                                        \
    }                                                                                 \
    break;
  switch (a.d) {
    NGPDE_FWD_CASE(16)
    NGPDE_FWD_CASE(32)
    NGPDE_FWD_CASE(64)
    NGPDE_FWD_CASE(128)
  }
#undef NGPDE_FWD_CASE
#undef NGPDE_FWD_LAUNCH
#undef NGPDE_FWD_LAUNCH2
  NGPDE_LAUNCH_CHECK("gcn_fused_fwd_kernel");
  return NGPDE_OK;
}

int32_t launch_fused_bwd(const FusedBwdArgs &a, hipStream_t stream) {
  const ngpde_graph *g = a.g;
  NGPDE_REQUIRE(g && g->has_norm, NGPDE_ERR_STATE, "GCN normalisation not set (call ngpde_graph_set_gcn_norm)");
  NGPDE_REQUIRE(fused_supported(a.d, a.d), NGPDE_ERR_UNSUPPORTED, "fused GCN path needs d in {16,32,64,128}, got %d", a.d);
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE((uint64_t)g->n_nodes * a.d * 4 < (1ull << 32), NGPDE_ERR_UNSUPPORTED,
                "fused GCN path addresses feature arrays with 32-bit byte offsets: n_nodes * d * 4 must be < 4 GiB");
  BwdK k;
  k.g_in = a.g_in; k.sched = g->by_s.sched; k.ent = g->by_s.ent; k.ell = g->by_s.ell;
  k.halo = g->by_s.halo; k.tile_info = g->by_s.tile_info; k.slots = g->by_s.slots; k.slot_w = g->by_s.slot_w;
  if (a.of && a.of->slots[1] && g->by_s.halo_ok && !no_halo_env()) {
    k.slots = a.of->slots[1]; k.sched = a.of->sched[1];
    if (k.slot_w) k.slot_w = a.of->slot_w[1];
  }
  k.self_loops = g->self_loops; k.n_tiles = fused_num_blocks(g->n_nodes); k.act = a.act;
  k.has_comb = a.has_comb ? 1 : 0; k.comb = to_dev(a.comb);
  k.store_t = a.store_t; k.store_v = a.store_v; k.v_scale = a.v_scale;
  k.do_dense = a.do_dense ? 1 : 0; k.z = a.z; k.saved_agg = a.saved_agg; k.wt = a.wt;
  k.mask = a.mask;
  NGPDE_REQUIRE(!a.mask || a.act == NGPDE_ACT_RELU, NGPDE_ERR_INVALID_ARGUMENT, "sign-bit masks carry relu' only");
  k.tape_late = a.has_comb ? 0 : 1;
  {
    static const int ord = [] { const char *e = std::getenv("NGPDE_BWD_ORDER"); return e ? atoi(e) : 1; }();
    k.order = ord;
    if (ord & 4) k.tape_late = 1;
  }
  k.g_out = a.g_out; k.slab_dw = a.slab_dw; k.slab_db = a.slab_db;
  NGPDE_STAMP_SET(k, k.n_tiles)
  const bool use_halo = g->by_s.halo_ok && !no_halo_env();
  NGPDE_REQUIRE(!a.pre || fused_prescaled_supported(g, a.d), NGPDE_ERR_UNSUPPORTED,
                "the pre-scaled form needs self loops and tiles that fit the LDS halo in both directions");
  const bool pair = fused_bwd_pairs(a.d);
  const dim3 grid(pair ? (k.n_tiles + 1) / 2 : k.n_tiles), block(pair ? 2 * kThreads : kThreads);
#define NGPDE_BWD_LAUNCH3(DD, AG, AA, HH, PP, SS)                                                                 \
  if (a.ev_start) hipExtLaunchKernelGGL((gcn_fused_bwd_kernel<DD, AG, AA, HH, PP, SS>), grid, block, 0, stream, a.ev_start, a.ev_stop, 0, \
                                        k.halo, k.slots, k.sched, k.g_in, k.slot_w, k.n_tiles, k);                    \
  else hipLaunchKernelGGL((gcn_fused_bwd_kernel<DD, AG, AA, HH, PP, SS>), grid, block, 0, stream, k.halo, k.slots, k.sched, k.g_in, k.slot_w, k.n_tiles, k);
#define NGPDE_BWD_LAUNCH2(DD, AG, AA, HH, SS)                                                                     \
  if (DD <= 64 && pair) { NGPDE_BWD_LAUNCH3(DD, AG, AA, HH, (DD <= 64), SS) } else { NGPDE_BWD_LAUNCH3(DD, AG, AA, HH, false, SS) }
#define NGPDE_BWD_LAUNCH(DD, AG, AA)                                                                              \
  if (Geo<DD>::HALO && a.pre) { NGPDE_BWD_LAUNCH2(DD, AG, AA, (AG && Geo<DD>::HALO), (Geo<DD>::HALO)) }            \
  else if (AG && Geo<DD>::HALO && use_halo) { NGPDE_BWD_LAUNCH2(DD, AG, AA, (AG && Geo<DD>::HALO), false) }        \
  else { NGPDE_BWD_LAUNCH2(DD, AG, AA, false, false) }
#define NGPDE_BWD_ACT(DD, AG)                                                         \
  switch (act_template(a.act)) {                                                      \
    case NGPDE_ACT_RELU: NGPDE_BWD_LAUNCH(DD, AG, NGPDE_ACT_RELU) break;              \
    case NGPDE_ACT_IDENTITY: NGPDE_BWD_LAUNCH(DD, AG, NGPDE_ACT_IDENTITY) break;      \
    default: NGPDE_BWD_LAUNCH(DD, AG, -1) break;                                      \
  }
#define NGPDE_BWD_CASE(DD)                                                            \
  case DD:                                                                            \
    if (a.aggregate) { NGPDE_BWD_ACT(DD, true) } else { NGPDE_BWD_ACT(DD, false) }    \
    break;
  switch (a.d) {
    NGPDE_BWD_CASE(16)
    NGPDE_BWD_CASE(32)
    NGPDE_BWD_CASE(64)
    NGPDE_BWD_CASE(128)
  }
#undef NGPDE_BWD_CASE
#undef NGPDE_BWD_ACT
#undef NGPDE_BWD_LAUNCH
#undef NGPDE_BWD_LAUNCH2
#undef NGPDE_BWD_LAUNCH3
  NGPDE_LAUNCH_CHECK("gcn_fused_bwd_kernel");
  return NGPDE_OK;
}

bool gat_fused_supported(const ngpde_graph *g, int heads, int c) {
  return g && g->has_norm && g->by_t.halo_ok && heads * c == 64 && (heads == 1 || heads == 2 || heads == 4) &&
         (uint64_t)g->n_nodes * 64 * 4 < (1ull << 32);
}

int32_t launch_gat_fused_fwd(const ngpde_graph *g, int heads, int c, float slope, const float *wx, const float *al, const float *ar,
                             float *out, float *alpha, hipStream_t stream) {
  NGPDE_REQUIRE(gat_fused_supported(g, heads, c), NGPDE_ERR_UNSUPPORTED, "fused GAT aggregation needs heads * c == 64, heads in {1,2,4}, tiles that fit the LDS halo");
  if (g->n_nodes == 0) return NGPDE_OK;
  GatK k;
  k.wx = wx; k.al = al; k.ar = ar; k.sched = g->by_t.sched; k.halo = g->by_t.halo; k.slots = g->by_t.slots;
  k.n_tiles = fused_num_blocks(g->n_nodes); k.heads = heads; k.slope = slope; k.out = out; k.alpha = alpha;
  hipLaunchKernelGGL(gat_fused_fwd_kernel, dim3(k.n_tiles), dim3(kThreads), 0, stream, k);
  NGPDE_LAUNCH_CHECK("gat_fused_fwd_kernel");
  return NGPDE_OK;
}

namespace {
__global__ void zero_bytes_kernel(unsigned *__restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
}  // namespace

// Zero a 4-byte-aligned device buffer with a KERNEL.  Used instead of hipMemsetAsync wherever a caller may capture the call
// into a HIP graph: a memset node was observed not to be ordered reliably against the neighbouring kernel nodes on replays.
int32_t launch_zero(void *ptr, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return NGPDE_OK;
  NGPDE_REQUIRE((reinterpret_cast<uintptr_t>(ptr) & 3) == 0 && bytes % 4 == 0, NGPDE_ERR_INVALID_ARGUMENT, "launch_zero: unaligned buffer");
  const size_t n = bytes / 4;
  hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, stream, (unsigned *)ptr, n);
  NGPDE_LAUNCH_CHECK("zero_bytes_kernel");
  return NGPDE_OK;
}

int32_t launch_reduce_slabs(const float *slab, int n_slabs, int len, int ct, float *out, hipStream_t stream) {
  if (len == 0) return NGPDE_OK;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((len + 63) / 64), dim3(1024), 0, stream, slab, n_slabs, len, ct, out);
  NGPDE_LAUNCH_CHECK("reduce_slabs_kernel");
  return NGPDE_OK;
}

}  // namespace ngpde

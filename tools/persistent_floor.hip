// Diagnostic: the floor of a PERSISTENT solver launch -- what does one phase cost when the 512 workgroups of the C2 solve stay
// resident and exchange their rows inside the launch instead of across 1205 kernel boundaries?  
//
// Each workgroup owns one 32-row tile of a [16384][64] fp32 array.  A phase = { wait until the tiles whose rows this tile's
// halo references have finished the previous phase -> gather the tile's 56 halo rows (own 32 + 3 rows of each of the 8
// surrounding tiles of a 32 x 16 tile grid, the shape of the C2 graph's cluster tiles) into LDS by LDS-DMA -> write the 32
// own rows of the other array -> publish }.  No arithmetic: this is the synchronisation + data-movement floor that a phase of
// a persistent kernel cannot go below, to be compared with the 1.87 us per node of a HIP-graph chain (launch_floor.hip) and
// with the 5.5-9.8 us the solver's launches take today.
//
// Modes (argv[1], default: all):
//   0  free-running (no synchronisation at all: WRONG results, the pure data-movement time)
//   1  neighbour flags: sc1 (write-through) row stores, every wave drains, one flag store per tile and phase; the consumer
//      polls the <= 9 flags of its halo tiles and gathers with sc1 loads (no fences)        [guide: Guideline 16, R1 + sc1 loads]
//   2  neighbour flags with plain stores + agent release fence, agent acquire fence + plain loads
//   3  grid barrier (8 sharded arrival counters, every workgroup polls all shards), sc1 stores / sc1 loads
//   4  grid barrier, plain stores + release fence, acquire fence + plain loads
//   6  as 1, with `pollers` (argv[7]) waves polling out of step; the first that sees every flag releases the others through LDS
//   7  inbox counters: every producer atomic-adds 1 to the (phase-parity) inbox word of each of its readers; a consumer polls ONE word
//   5  tagged packets, no flags at all: every exchanged dword travels as an 8-byte packet {value, phase tag} (sc1 stores, two
//      dwordx4 per lane = four packets); the consumer's lanes poll THEIR OWN packets of the foreign halo rows (sc1 loads) until
//      all four tags carry the producer's phase, then put the values into LDS.  One fabric traversal per phase instead of
//      flag-then-rows; needs a symmetric halo relation (a tile that reads a neighbour's rows is read by it) for the
//      write-after-read safety of the two ping-pong buffers.  Own rows stay in LDS (foreign rows only).
//   8  tagged QUADS, no flags, no drain (round 5): a row travels as 22 sixteen-byte quads {f, f, f, phase tag}
//      (lane q of the row's 16 lanes stores {its floats 0..2, tag}; lanes 0, 3, .. 15 also store the three neighbouring lanes'
//      fourth floats as quad 16 + q / 3) -- one dwordx4 store per quad, 352 B per row in a 384-B stride.  The consumer's gather IS
//      the poll: LDS-DMA of the quads, s_waitcnt, every lane checks the tag of the quad it asked for IN LDS and asks again only
//      for the stale ones (exec-masked DMA).  argv[6] = s_sleep rounds between polling rounds.
//   9  tagged quads + flags WITHOUT the drain: the producer stores its rows, meets at a barrier (stores issued, not acknowledged),
//      one lane stores the flag; the consumer polls the flags as in mode 1, gathers, validates the tags and re-gathers stale quads.
// argv: [mode or -1] [phases] [work in 10 ns ticks] [foreign rows only 0/1] [poll without s_sleep 0/1] [mode 5: s_sleep between polls]
// Every payload word carries (phase, row), every gathered word is checked, and every spin is bounded (abort word + timeout).
// build + run:  hipcc -O2 --offload-arch=gfx950 tools/persistent_floor.hip -o /tmp/persistent_floor && /tmp/persistent_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kRows = 32, kD = 64, kLpr = kD / 4, kThreads = 512, kHalo = 56, kNbr = 9;
constexpr int kGridX = 32, kGridY = 16, kTiles = kGridX * kGridY;
constexpr int kQStride = 24;   // quads per row of the tagged-quad buffers (22 used): 384 B = three 128-B lines

typedef unsigned gu32 __attribute__((address_space(1)));
typedef float f4v __attribute__((ext_vector_type(4)));

struct Params {
  float *buf[2];
  float *pbuf[2];       // mode 5: [rows][64][2] {value, tag} packets
  unsigned *flags;      // [kTiles] last finished phase + 1, one 128-B line each
  unsigned *shards;     // [8] arrival counters, one 128-B line each
  unsigned *abort_word; // != 0: somebody timed out
  unsigned *errors;     // mismatching payload words
  const int *halo;      // [kTiles][kHalo] row ids
  const int *nbr;       // [kTiles][kNbr] tiles to wait for (own tile included: the write-after-read hazard on the ping-pong arrays)
  int phases, mode, check;
  int work_ticks;    // simulated arithmetic between the gather and the stores, in 10 ns ticks (s_memrealtime)
  int foreign_only;  // gather only the 24 rows of OTHER tiles (a persistent kernel keeps its own rows on chip)
  int no_sleep;
  int poll_sleep;    // mode 5: s_sleep rounds between polls; mode 6: start offset of polling wave k = k * poll_sleep sleeps
  int pollers;       // mode 6: polling waves
};

__device__ __forceinline__ int xcd_tile(int b, int nb) {
  const int x = b % 8, k = b / 8, q = nb / 8, r = nb % 8;
  return x * q + min(x, r) + k;
}

__device__ __forceinline__ unsigned payload(int phase, int row, int c) { return ((unsigned)phase << 20) ^ ((unsigned)row << 6) ^ (unsigned)c; }

__device__ __forceinline__ void store_sc1(float *p, f4v v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// bounded spin: returns false when the abort word is set or ~40 ms have passed
__device__ __forceinline__ bool spin_ok(unsigned long long t0, unsigned *abort_word) {
  if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
  if (__builtin_amdgcn_s_memrealtime() - t0 > 4000000ull) {   // 100 MHz ticks
    __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
  }
  return true;
}

__global__ __launch_bounds__(kThreads, 4) void persistent_kernel(const Params p) {
  __shared__ __attribute__((aligned(16))) float lds[(kHalo + 8) * kD];
  __shared__ __attribute__((aligned(16))) unsigned tag_lds[2 * kHalo * 16 * 4];   // modes 8 / 9
  __shared__ int s_ok, s_go;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = tid / kLpr, q = tid % kLpr;
  const int tile = xcd_tile(blockIdx.x, kTiles);
  const bool sc1 = (p.mode == 1 || p.mode == 3 || p.mode == 6 || p.mode == 7), fences = (p.mode == 2 || p.mode == 4);
  // metadata stays in registers for the whole launch: halo row ids of this thread's DMA slots, the flags this lane polls
  int hrow[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) hrow[k] = p.halo[tile * kHalo + min(grp + 32 * k, kHalo - 1)];
  const int my_nbr = (lane < kNbr) ? p.nbr[tile * kNbr + lane] : tile;
  unsigned bad = 0;
  if (tid == 0) { s_ok = 1; s_go = 0; }
  __syncthreads();
  if (p.mode == 5) {
    // ---- tagged packets: no flags, no drain, no publish ----
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    for (int ph = 1; ph <= p.phases; ++ph) {
      const unsigned *src = reinterpret_cast<const unsigned *>(p.pbuf[(ph + 1) & 1]);
      unsigned *dst = reinterpret_cast<unsigned *>(p.pbuf[ph & 1]);
      if (ph > 1) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool done[2] = {false, false};
        const unsigned want = (unsigned)(ph - 1);
        bool ok = true;
        for (unsigned it = 1;; ++it) {
          bool all_done = true;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int hh = grp + 32 * k;
            const bool mine = hh >= kRows && hh < kHalo;         // foreign rows only: slots 32 .. 55
            if (mine && !done[k]) {
              const unsigned *g = src + (size_t)hrow[k] * (2 * kD) + q * 8;
              u4v a, b;
              asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                           : "=&v"(a), "=&v"(b) : "v"(g) : "memory");
              if (a[1] == want && a[3] == want && b[1] == want && b[3] == want) {
                done[k] = true;
                f4v v = {__uint_as_float(a[0]), __uint_as_float(a[2]), __uint_as_float(b[0]), __uint_as_float(b[2])};
                reinterpret_cast<f4v *>(lds)[hh * kLpr + q] = v;
              } else {
                all_done = false;
              }
            }
          }
          if (!__any((int)!all_done)) break;
          if ((it & 255u) == 0 && !spin_ok(t0, p.abort_word)) { ok = false; break; }
          for (int sl = 0; sl < p.poll_sleep; ++sl) __builtin_amdgcn_s_sleep(1);   // back-off between polling rounds (64 clocks each)
        }
        if (!ok) s_ok = 0;
        __syncthreads();
        if (!s_ok) break;
        if (p.check) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int hh = grp + 32 * k;
            if (hh >= kRows && hh < kHalo) {
              const f4v v = reinterpret_cast<const f4v *>(lds)[hh * kLpr + q];
              const int row = p.halo[tile * kHalo + hh];
#pragma unroll
              for (int c = 0; c < 4; ++c) bad += (__float_as_uint(v[c]) != payload(ph - 1, row, 4 * q + c));
            }
          }
        }
      }
      if (p.work_ticks > 0) {
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - w0 < (unsigned long long)p.work_ticks) __builtin_amdgcn_s_sleep(2);
      }
      {
        const int row = tile * kRows + grp;
        u4v a = {payload(ph, row, 4 * q), (unsigned)ph, payload(ph, row, 4 * q + 1), (unsigned)ph};
        u4v b = {payload(ph, row, 4 * q + 2), (unsigned)ph, payload(ph, row, 4 * q + 3), (unsigned)ph};
        unsigned *d = dst + (size_t)row * (2 * kD) + q * 8;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" ::"v"(d), "v"(a), "v"(b) : "memory");
      }
      __syncthreads();   // everybody is done with the LDS halo before the next phase refills it
    }
    if (bad) atomicAdd(p.errors, bad);
    return;
  }
  if (p.mode == 8 || p.mode == 9) {
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    // LDS image of the gathered quads: [main | extra][slot][16] -- a wave's DMA lands lane-linearly, so the main quads of its four
    // slots are 64 consecutive quads; the extra region uses the first 6 of every 16 (prototype: not packed)
    u4v *tl = reinterpret_cast<u4v *>(tag_lds);
    volatile __attribute__((address_space(3))) unsigned *tag3 = (volatile __attribute__((address_space(3))) unsigned *)tag_lds;
    const int hh = grp + kRows;                      // this thread's foreign slot (32 .. 55 for waves 0 .. 5)
    const bool have = hh < kHalo;
    const int wbase = (kRows + 4 * __builtin_amdgcn_readfirstlane(wave)) * 16;   // first quad of the wave's four slots
    const int frow = p.halo[tile * kHalo + min(hh, kHalo - 1)];
    for (int ph = 1; ph <= p.phases; ++ph) {
      const unsigned *src = reinterpret_cast<const unsigned *>(p.pbuf[(ph + 1) & 1]);
      unsigned *dst = reinterpret_cast<unsigned *>(p.pbuf[ph & 1]);
      if (ph > 1) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned want = (unsigned)(ph - 1);
        bool ok = true;
        if (p.mode == 9) {
          if (wave == 0) {
            for (;;) {
              const unsigned f = __hip_atomic_load(p.flags + 32 * my_nbr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (__all((int)(f >= want))) break;
              if (!spin_ok(t0, p.abort_word)) { ok = false; break; }
              if (!p.no_sleep) __builtin_amdgcn_s_sleep(1);
            }
            if (!ok && lane == 0) s_ok = 0;
          }
          __syncthreads();
          if (!s_ok) break;
        }
        bool need_m = have, need_x = have && q < 6;
        const char *gm = reinterpret_cast<const char *>(src) + (size_t)frow * (kQStride * 16) + q * 16;
        const char *gx = gm + 16 * 16;
        for (unsigned it = 1;; ++it) {
          if (need_m)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gm,
                                             (__attribute__((address_space(3))) void *)(tl + wbase), 16, 0, 16);
          if (need_x)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gx,
                                             (__attribute__((address_space(3))) void *)(tl + kHalo * 16 + wbase), 16, 0, 16);
          __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0) (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt / lgkmcnt left at max)
          if (need_m) need_m = (tag3[(hh * 16 + q) * 4 + 3] != want);
          if (need_x) need_x = (tag3[(kHalo * 16 + hh * 16 + q) * 4 + 3] != want);
          if (!__any((int)(need_m || need_x))) break;
          if ((it & 63u) == 0 && !spin_ok(t0, p.abort_word)) { ok = false; break; }
          for (int sl = 0; sl < p.poll_sleep; ++sl) __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) s_ok = 0;
        __syncthreads();
        if (!s_ok) break;
        if (p.check && have) {
          const u4v m = tl[hh * 16 + q];
#pragma unroll
          for (int c = 0; c < 3; ++c) bad += (m[c] != payload(ph - 1, frow, 4 * q + c));
          if (q < 6) {
            const u4v x = tl[kHalo * 16 + hh * 16 + q];
#pragma unroll
            for (int c = 0; c < 3; ++c)
              if (3 * q + c < 16) bad += (x[c] != payload(ph - 1, frow, 4 * (3 * q + c) + 3));
          }
        }
      }
      if (p.work_ticks > 0) {
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - w0 < (unsigned long long)p.work_ticks) __builtin_amdgcn_s_sleep(2);
      }
      {
        const int row = tile * kRows + grp;
        unsigned *d = dst + (size_t)row * (kQStride * 4);
        u4v m = {payload(ph, row, 4 * q), payload(ph, row, 4 * q + 1), payload(ph, row, 4 * q + 2), (unsigned)ph};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(d + 4 * q), "v"(m) : "memory");
        if (q % 3 == 0) {   // the real kernel gets the neighbours' fourth floats by two DPP row shifts
          u4v x = {payload(ph, row, 4 * q + 3), q + 1 < 16 ? payload(ph, row, 4 * (q + 1) + 3) : 0u,
                   q + 2 < 16 ? payload(ph, row, 4 * (q + 2) + 3) : 0u, (unsigned)ph};
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(d + 4 * (16 + q / 3)), "v"(x) : "memory");
        }
      }
      if (p.mode == 9) {
        __builtin_amdgcn_s_barrier();    // every wave has ISSUED its stores; nobody waits for their acknowledgement
        if (tid == 0) __hip_atomic_store(p.flags + 32 * tile, (unsigned)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();   // everybody is done with the LDS image before the next phase refills it
    }
    if (bad) atomicAdd(p.errors, bad);
    return;
  }
  for (int ph = 1; ph <= p.phases; ++ph) {
    const float *src = p.buf[(ph + 1) & 1];
    float *dst = p.buf[ph & 1];
    // ---- wait for the producers of this phase's halo rows (they finished phase ph - 1) ----
    if (ph > 1 && p.mode == 7) {
      // inbox counters: one word per tile and phase parity, every producer adds 1 to the inbox of each tile that reads it; the
      // consumer polls ONE word (its own inbox of the previous phase's parity) instead of one flag per neighbour
      if (wave == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned need = (unsigned)kNbr * (unsigned)(ph / 2);      // phases 1 .. ph - 1 with the parity of ph - 1: ph / 2 of them
        bool ok = true;
        for (;;) {
          const unsigned f = __hip_atomic_load(p.flags + 32 * tile + ((ph - 1) & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (f >= need) break;
          if (!spin_ok(t0, p.abort_word)) { ok = false; break; }
          if (!p.no_sleep) __builtin_amdgcn_s_sleep(1);
        }
        if (!ok && lane == 0) s_ok = 0;
      }
      __syncthreads();
      if (!s_ok) break;
    } else if (ph > 1 && p.mode == 6) {
      // several polling waves, out of step with each other: the first one that sees every flag releases the others through LDS
      // (a poll round is a fabric round trip; with one poller a flag that lands right after a poll was issued waits a whole round)
      if (wave < p.pollers) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int sl = 0; sl < wave * p.poll_sleep; ++sl) __builtin_amdgcn_s_sleep(1);
        for (;;) {
          if (*(volatile int *)&s_go == ph) break;
          const unsigned f = __hip_atomic_load(p.flags + 32 * my_nbr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__all((int)(f >= (unsigned)(ph - 1)))) { if (lane == 0) *(volatile int *)&s_go = ph; break; }
          if (!spin_ok(t0, p.abort_word)) { if (lane == 0) { s_ok = 0; *(volatile int *)&s_go = ph; } break; }
        }
      }
      __syncthreads();
      if (!s_ok) break;
    } else if (ph > 1 && p.mode != 0) {
      if (wave == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool ok = true;
        if (p.mode == 1 || p.mode == 2) {
          for (;;) {
            const unsigned f = __hip_atomic_load(p.flags + 32 * my_nbr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(f >= (unsigned)(ph - 1)))) break;
            if (!spin_ok(t0, p.abort_word)) { ok = false; break; }
            if (!p.no_sleep) __builtin_amdgcn_s_sleep(1);
          }
        } else {
          const unsigned target = (unsigned)(ph - 1) * (kTiles / 8);
          for (;;) {
            const unsigned f = __hip_atomic_load(p.shards + 32 * (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(f >= target))) break;
            if (!spin_ok(t0, p.abort_word)) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        if (fences) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!ok && lane == 0) s_ok = 0;
      }
      __syncthreads();
      if (!s_ok) break;   // uniform: every wave of the workgroup leaves
    }
    // ---- gather the halo rows into LDS (LDS-DMA, 16 B per lane, a wave's four 16-lane groups = four consecutive slots) ----
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int hh = grp + 32 * k;
      if (hh < kHalo && !(p.foreign_only && hh < kRows)) {   // wave-uniform (kHalo is a multiple of 4)
        const char *g = reinterpret_cast<const char *>(src) + (size_t)hrow[k] * (kD * 4) + q * 16;
        __attribute__((address_space(3))) void *l = (__attribute__((address_space(3))) void *)(reinterpret_cast<f4v *>(lds) + hh * kLpr + q);
        if (sc1) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, l, 16, 0, 16);
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, l, 16, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.check && ph > 1 && p.mode != 0) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int hh = grp + 32 * k;
        if (hh < kHalo && !(p.foreign_only && hh < kRows)) {
          const f4v v = reinterpret_cast<const f4v *>(lds)[hh * kLpr + q];
          const int row = p.halo[tile * kHalo + hh];
#pragma unroll
          for (int c = 0; c < 4; ++c) bad += (__float_as_uint(v[c]) != payload(ph - 1, row, 4 * q + c));
        }
      }
    }
    if (p.work_ticks > 0) {   // stand-in for LDS aggregation + MFMA + epilogue
      const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - w0 < (unsigned long long)p.work_ticks) __builtin_amdgcn_s_sleep(2);
    }
    // ---- write the own rows of the other array ----
    {
      const int row = tile * kRows + grp;   // 32 groups of 16 lanes = 32 rows x 256 B
      f4v v;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = __uint_as_float(payload(ph, row, 4 * q + c));
      float *d = dst + (size_t)row * kD + 4 * q;
      if (sc1) store_sc1(d, v);
      else *reinterpret_cast<f4v *>(d) = v;
    }
    // ---- publish ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains
    __syncthreads();                                    // (also: everybody is done reading the LDS halo)
    if (p.mode == 7 && tid < kNbr)
      __hip_atomic_fetch_add(p.flags + 32 * p.nbr[tile * kNbr + tid] + (ph & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p.mode != 0 && p.mode != 7 && tid == 0) {
      if (fences) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (p.mode == 1 || p.mode == 2 || p.mode == 6) __hip_atomic_store(p.flags + 32 * tile, (unsigned)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(p.shards + 32 * (blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (bad) atomicAdd(p.errors, bad);
}

int main(int argc, char **argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  const int phases = argc > 2 ? atoi(argv[2]) : 1200;
  const int work = argc > 3 ? atoi(argv[3]) : 0, foreign = argc > 4 ? atoi(argv[4]) : 0, no_sleep = argc > 5 ? atoi(argv[5]) : 0;
  const int poll_sleep = argc > 6 ? atoi(argv[6]) : 0, pollers = argc > 7 ? atoi(argv[7]) : 2;
  const size_t elems = (size_t)kTiles * kRows * kD;
  Params p{};
  // FLOOR_MEM: how the flag lines (bit 0) and the exchanged arrays (bit 1) are allocated -- 0 / unset: hipMalloc; FLOOR_MEMKIND
  // chooses the flavour for the selected ones: "fine" = hipDeviceMallocFinegrained, "uncached" = hipDeviceMallocUncached
  const int mem_sel = getenv("FLOOR_MEM") ? atoi(getenv("FLOOR_MEM")) : 0;
  const char *mem_kind = getenv("FLOOR_MEMKIND") ? getenv("FLOOR_MEMKIND") : "fine";
  const unsigned mem_flag = (mem_kind[0] == 'u') ? hipDeviceMallocUncached : hipDeviceMallocFinegrained;
  auto alloc = [&](void **ptr, size_t bytes, bool special) -> hipError_t {
    return special ? hipExtMallocWithFlags(ptr, bytes, mem_flag) : hipMalloc(ptr, bytes);
  };
  CK(alloc((void **)&p.buf[0], elems * 4, mem_sel & 2));
  CK(alloc((void **)&p.buf[1], elems * 4, mem_sel & 2));
  CK(hipMalloc(&p.pbuf[0], elems * 8));
  CK(hipMalloc(&p.pbuf[1], elems * 8));
  CK(alloc((void **)&p.flags, kTiles * 128, mem_sel & 1));
  CK(hipMalloc(&p.shards, 8 * 128));
  CK(hipMalloc(&p.abort_word, 128));
  CK(hipMalloc(&p.errors, 128));
  std::vector<int> halo(kTiles * kHalo), nbr(kTiles * kNbr);
  for (int t = 0; t < kTiles; ++t) {
    const int x = t % kGridX, y = t / kGridX;
    int h = 0, n = 0;
    for (int r = 0; r < kRows; ++r) halo[t * kHalo + h++] = t * kRows + r;
    nbr[t * kNbr + n++] = t;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        if (!dx && !dy) continue;
        const int u = ((y + dy + kGridY) % kGridY) * kGridX + (x + dx + kGridX) % kGridX;
        nbr[t * kNbr + n++] = u;
        for (int r = 0; r < 3; ++r) halo[t * kHalo + h++] = u * kRows + (7 * (dx + 1) + 3 * (dy + 1) + 11 * r) % kRows;
      }
  }
  int *d_halo, *d_nbr;
  CK(hipMalloc(&d_halo, halo.size() * 4));
  CK(hipMalloc(&d_nbr, nbr.size() * 4));
  CK(hipMemcpy(d_halo, halo.data(), halo.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_nbr, nbr.data(), nbr.size() * 4, hipMemcpyHostToDevice));
  p.halo = d_halo; p.nbr = d_nbr; p.phases = phases; p.work_ticks = work; p.foreign_only = foreign; p.no_sleep = no_sleep; p.poll_sleep = poll_sleep; p.pollers = pollers;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char *names[] = {"free-running (no sync, data movement only)", "neighbour flags, sc1 stores + sc1 loads",
                         "neighbour flags, plain stores + release / acquire fences", "grid barrier (8 shards), sc1 stores + sc1 loads",
                         "grid barrier (8 shards), plain stores + release / acquire fences",
                         "tagged 8-byte packets {value, phase}, no flags (foreign rows only)",
                         "neighbour flags, sc1, several polling waves out of step",
                         "inbox counters (one polled word per tile, producers atomic-add), sc1",
                         "tagged 16-byte quads {f, f, f, phase}: the LDS-DMA gather is the poll, no flags, no drain",
                         "tagged quads + flags without the drain (stale quads re-gathered)"};
  for (int mode = 0; mode < 10; ++mode) {
    if (only >= 0 && mode != only) continue;
    for (int check = 1; check >= 0; --check) {
      p.mode = mode; p.check = check;
      float best = 1e30f;
      unsigned err = 0, ab = 0;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(p.buf[0], 0, elems * 4, s));
        CK(hipMemsetAsync(p.buf[1], 0, elems * 4, s));
        CK(hipMemsetAsync(p.pbuf[0], 0, elems * 8, s));
        CK(hipMemsetAsync(p.pbuf[1], 0, elems * 8, s));
        CK(hipMemsetAsync(p.flags, 0, kTiles * 128, s));
        CK(hipMemsetAsync(p.shards, 0, 8 * 128, s));
        CK(hipMemsetAsync(p.abort_word, 0, 128, s));
        CK(hipMemsetAsync(p.errors, 0, 128, s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(persistent_kernel, dim3(kTiles), dim3(kThreads), 0, s, p);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned e = 0, a = 0;
        CK(hipMemcpy(&e, p.errors, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&a, p.abort_word, 4, hipMemcpyDeviceToHost));
        err += e; ab += a;
        if (rep > 0 && ms < best) best = ms;
        if (a) break;
      }
      printf("{\"mode\": %d, \"what\": \"%s\", \"grid\": %d, \"block\": %d, \"phases\": %d, \"payload_check\": %d, "
             "\"work_us\": %.2f, \"foreign_rows_only\": %d, \"no_sleep\": %d, \"poll_sleep\": %d, \"us_per_phase\": %.3f, \"bad_words\": %u, \"aborted\": %u}\n",
             mode, names[mode], kTiles, kThreads, phases, check, work * 0.01, foreign, no_sleep, poll_sleep, best * 1000.f / phases, err, ab);
      fflush(stdout);
    }
  }
  return 0;
}

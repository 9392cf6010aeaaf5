// Device-side helpers shared by the two persistent denominator kernels (chain_den_persist.hip, chain_den_persist2.hip):
// the team control block, agent-scope loads / stores, the exchange words and their poll, LDS-DMA, wave reductions.
#pragma once
#include "den_persist.h"
#include "persist_guard.h"
#include "step_graph.h"

namespace pk2 {

constexpr unsigned kRingSentinel = 0x7fc0dead;       // a NaN payload no arithmetic produces
constexpr int kMaxTeams = 8;                         // teams per XCD the control block has room for
constexpr int kMaxTasks = 64;
constexpr long long kDenSpinTicks = 1000LL * 1000 * 100;   // 1 s of the 100 MHz wall clock

#ifdef PK2_DP_PROFILE
// Phase timers of rank 0 / thread 0 (shader clocks, averaged over the frames of a call) and the wall-clock timeline of one
// frame on every rank.  The counters live in `dp_` (DP_T0 declares it; helper functions receive it by reference).
struct DpTimers { long long last; unsigned long long acc[10]; int tl_on, tl_dir, tl_rank; };
static __device__ unsigned long long g_dp_prof[2][10];
#define DP_T0() DpTimers dp_; dp_.last = clock64(); for (int k_ = 0; k_ < 10; ++k_) dp_.acc[k_] = 0
#define DP_T(k) do { const long long n_ = clock64(); dp_.acc[k] += (unsigned long long)(n_ - dp_.last); dp_.last = n_; } while (0)
#define DP_FLUSH(dir) do { if (rank == 0 && threadIdx.x == 0) for (int k_ = 0; k_ < 10; ++k_) atomicAdd(&g_dp_prof[dir][k_], dp_.acc[k_]); } while (0)
static __device__ long long g_dp_tl[2][kPR][8];     // wall-clock (10 ns) timeline of frame 100 of the first sequence, per rank
#define DP_TL(dir, k) do { if (g == 0 && t == 100 && threadIdx.x == 0) g_dp_tl[dir][rank][k] = wall_clock64(); } while (0)
#define DP_TLSET(dir) do { dp_.tl_on = (g == 0 && t == 100) ? 1 : 0; dp_.tl_dir = dir; dp_.tl_rank = rank; } while (0)
#define DP_TLF(k) do { if (dp_.tl_on && threadIdx.x == 0) g_dp_tl[dp_.tl_dir][dp_.tl_rank][k] = wall_clock64(); } while (0)
static __global__ void dp_prof_print(int frames, int form) {
  for (int dir = 0; dir < 2; ++dir) {
    long long t0 = g_dp_tl[dir][0][0];
    for (int r = 0; r < kPR; ++r) t0 = g_dp_tl[dir][r][0] < t0 ? g_dp_tl[dir][r][0] : t0;
    printf("timeline %s frame 100 (10 ns ticks after the first rank entered the frame): rank: enter, partials valid, chunk 0 ready (v1: table), after pass A (v1: after barrier), arcs done, epilogue done, published\n", dir ? "bwd" : "fwd");
    for (int r = 0; r < kPR; ++r)
      printf("  %2d: %lld %lld %lld %lld %lld %lld %lld\n", r, g_dp_tl[dir][r][0] - t0, g_dp_tl[dir][r][1] - t0, g_dp_tl[dir][r][2] - t0,
             g_dp_tl[dir][r][3] - t0, g_dp_tl[dir][r][4] - t0, g_dp_tl[dir][r][5] - t0, g_dp_tl[dir][r][6] - t0);
  }
  for (int dir = 0; dir < 2; ++dir) {
    if (form == 2)
      printf("den_persist2 %s rank 0 thread 0, shader clocks per frame over %d frames: poll + sync + stage + copies issued %llu | chunk 0 landed + barrier %llu | pass A %llu | chunk 1 landed + barrier %llu | pass B (+ streamed) %llu | barrier + epilogue %llu | wait stores + block sum + publish %llu | history + prefetch %llu | streamed segment 0 %llu | streamed segments 1.. %llu\n",
             dir ? "bwd" : "fwd", frames, g_dp_prof[dir][0] / frames, g_dp_prof[dir][1] / frames, g_dp_prof[dir][2] / frames,
             g_dp_prof[dir][3] / frames, g_dp_prof[dir][4] / frames, g_dp_prof[dir][5] / frames, g_dp_prof[dir][6] / frames, g_dp_prof[dir][7] / frames, g_dp_prof[dir][8] / frames, g_dp_prof[dir][9] / frames);
    else
      printf("den_persist %s rank 0 thread 0, shader clocks per frame over %d frames: exchange (poll + table) %llu | barrier %llu | arcs %llu | barrier+fixup+barrier %llu | epilogue %llu | wait stores + block sum + publish %llu | prefetch %llu\n",
             dir ? "bwd" : "fwd", frames, g_dp_prof[dir][0] / frames, g_dp_prof[dir][1] / frames, g_dp_prof[dir][2] / frames,
             g_dp_prof[dir][3] / frames, g_dp_prof[dir][4] / frames, g_dp_prof[dir][5] / frames, g_dp_prof[dir][6] / frames);
    for (int k = 0; k < 10; ++k) g_dp_prof[dir][k] = 0;
  }
}
#else
struct DpTimers { };
#define DP_T0() DpTimers dp_; (void)dp_
#define DP_T(k) do { } while (0)
#define DP_FLUSH(dir) do { } while (0)
#define DP_TL(dir, k) do { } while (0)
#define DP_TLSET(dir) do { } while (0)
#define DP_TLF(k) do { } while (0)
#endif

struct DenPersistCtl {
  unsigned arrive[8];     // workgroups arrived per XCD: team = slot / kPR, rank = slot % kPR
  unsigned next_task;     // queue head
  unsigned abort;
  unsigned done;          // recursions completed
  unsigned pad[5];
  struct Team {
    unsigned task[kMaxTasks + 1];   // task[i] = 1 + index of the i-th recursion of this team (written by its rank 0)
    unsigned bar;                   // team barrier: arrivals
    unsigned pad[62];
  } team[8][kMaxTeams];
};

struct DenPersistParams {
  DenParams d;
  DevPersist fwd, bwd;
  const float* xv;        // [G][Tmax][V]
  float* ring;            // [8 * kMaxTeams][2][rpad]
  float* pring;           // [8 * kMaxTeams][3][kPR][kPWords]
  int rpad;               // floats per ring slot and of the LDS table
  int cap;                // LDS row buffers
  int ntasks;
  short task_seq[kMaxTasks];
  unsigned char task_dir[kMaxTasks];
};

// The parameter block is read through the constant address space: uniform fields become scalar loads (SGPRs), and no
// store of the kernel can be assumed to clobber them.
typedef __attribute__((address_space(4))) const DenPersistParams CParams;
typedef __attribute__((address_space(4))) const DenParams CDenParams;

// Arguments of a (non-inlined) device function arrive in VGPRs: the compiler must treat them -- and everything loaded
// through them -- as divergent.  Passing the wave-uniform ones through readfirstlane turns the parameter block's fields
// into scalar loads and their pointer arithmetic into SALU work (dozens of VGPRs next to the 96 of the arcs).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* uni(T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ CParams* uni(CParams* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (CParams*)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ unsigned den_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0x7;
}
// Pointers that arrive through memory (the parameter block) or as function arguments are generic: their accesses would be
// FLAT instructions, which count on lgkmcnt as well as vmcnt -- every LDS read of the arc loop would then wait for the
// global prefetches in flight.  The hot pointers are therefore cast to the global address space once.
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) const float cgfloat;
typedef __attribute__((address_space(1))) unsigned gunsigned;
__device__ __forceinline__ gfloat* G(float* p) { return (gfloat*)p; }
__device__ __forceinline__ cgfloat* G(const float* p) { return (cgfloat*)p; }
__device__ __forceinline__ gunsigned* G(unsigned* p) { return (gunsigned*)p; }
__device__ __forceinline__ float ld_agent(cgfloat* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(gfloat* p, float v) {      // the exchange words (and their resets)
#if PK2_DP_STOREMODE == 1
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ unsigned ld_agent_u(gunsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool is_sentinel(float v) { return __float_as_uint(v) == kRingSentinel; }
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// How the state vectors travel (build-time switches for A/B runs).  Loads: agent-scope (sc1, L1-bypassing); plain loads
// behind `buffer_inv sc0` were no faster (11.77 vs 11.75 ms per call) and keep stale lines.  Stores: PLAIN since round 4.
// A recursion's team lives on one XCD, whose L2 is the coherence point of its 32 CUs: the write-through vector L1
// forwards a plain store there and vmcnt acknowledges it there, so "slice stored -> s_waitcnt vmcnt(0) -> barrier -> word"
// is a release inside the XCD; an agent-scope store is additionally written through the L2 to the fabric on this
// multi-XCD part and its acknowledgement waits for that.  tools/ubench/handoff_plain.hip hammers exactly this protocol
// (32 workgroups of one XCD, uneven load): 0 stale words in 10^11 reads, for plain data + agent-scope flag and for plain
// data + plain flag alike; round 2's "the announcing word overtakes the data on small graphs" did not reproduce on the
// round-3/4 kernels (3 x 78 chain tests, small graphs included).  -DPK2_DP_STOREMODE=1 restores the agent-scope stores.
#ifndef PK2_DP_LOADMODE
#define PK2_DP_LOADMODE 0      // 0: agent-scope loads; 1: buffer_inv sc0 + plain loads; 2: buffer_inv sc1 + plain loads
#endif
#ifndef PK2_DP_STOREMODE
#define PK2_DP_STOREMODE 0     // 0: plain stores; 1: agent-scope stores
#endif
__device__ __forceinline__ void invalidate_l1() {
#if PK2_DP_LOADMODE == 1
  asm volatile("buffer_inv sc0" ::: "memory");
#elif PK2_DP_LOADMODE == 2
  asm volatile("buffer_inv sc1" ::: "memory");
#endif
}
__device__ __forceinline__ float ring_load(cgfloat* p) {
#if PK2_DP_LOADMODE == 0
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return *p;
#endif
}
__device__ __forceinline__ void ring_store(gfloat* p, float v) {
#if PK2_DP_STOREMODE == 1
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}

// Sum over the 64 lanes with DPP moves (full-rate VALU; __shfl_xor is a ds_bpermute round trip per step), the total
// broadcast from lane 63: the same bits in every lane.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_f<0x111, 0xf>(v);      // row_shr:1
  v += dpp_f<0x112, 0xf>(v);      // row_shr:2
  v += dpp_f<0x114, 0xf>(v);      // row_shr:4
  v += dpp_f<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row holds the row's sum
  v += dpp_f<0x142, 0xa>(v);      // row_bcast15 -> rows 1 and 3 add the sum of the row before
  v += dpp_f<0x143, 0xc>(v);      // row_bcast31 -> rows 2 and 3 add the sum of rows 0..1
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// A poll that keeps failing reads the wall clock every 256 rounds; after 1 s (or when somebody else gave up) it raises
// the abort flag.
struct Spin {
  DenPersistCtl* ctl;
  long long t0;
  unsigned n;
  __device__ __forceinline__ explicit Spin(DenPersistCtl* c) : ctl(c), t0(0), n(0) {}
  __device__ __forceinline__ bool expired() {
    if ((++n & 255u) != 0u) return false;
    const long long now = wall_clock64();
    if (t0 == 0) t0 = now;
    if (now - t0 > kDenSpinTicks || ld_agent_u(G(&ctl->abort))) {
      __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
    return false;
  }
};

// All kPR workgroups of the team have made their earlier stores visible and arrived `nbar` times.
__device__ __forceinline__ bool team_barrier(DenPersistCtl* ctl, DenPersistCtl::Team* team, unsigned* nbar, int* s_abort) {
  wait_stores();
  __syncthreads();
  const unsigned target = (unsigned)kPR * ++*nbar;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&team->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Spin spin(ctl);
    while (ld_agent_u(G(&team->bar)) < target) {
      if (spin.expired()) { *s_abort = 1; break; }
    }
  }
  __syncthreads();
  return *s_abort == 0;
}

// LDS-DMA (gfx950 global_load_lds_dwordx4): 256 consecutive floats of global memory -> 256 consecutive floats of LDS per
// wave instruction, no VGPR round trip; asynchronous (vmcnt).  Both addresses 16-byte aligned.
__device__ __forceinline__ void dma256(cgfloat* gsrc_lane, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, PK2_DP_LOADMODE == 0 ? 16 : 0);
}

// Exchange.  Every rank owns four words per frame slot: {partial sum 0, partial sum 1, "slice ready", -}.  A rank
// publishes a word AFTER the stores it announces have reached L2, so "the word of all kPR ranks is valid" means the data
// is complete.  ONE wave per workgroup polls (lane l: rank l % 32, word first + l / 32; 256 waves polling the same cache
// lines made a frame wait ~9 us for a value that was already there), the others wait at the barrier that follows; the
// sums over the ranks travel through LDS (tot[0], tot[1]: the same bits in every thread).
constexpr int kPWords = 4;
template <typename LDS>
__device__ __forceinline__ void poll_words(cgfloat* ps, int first, int nwords, Spin& spin, const LDS& L) {
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    const bool mine = (lane >> 5) < nwords;
    cgfloat* src = ps + (lane & (kPR - 1)) * kPWords + first + (lane >> 5);
    float pv = mine ? ld_agent(src) : 0.f;
    bool ok = true;
    while (__ballot(mine && is_sentinel(pv)) != 0ull) {
      if (spin.expired()) { ok = false; break; }
      if (mine && is_sentinel(pv)) pv = ld_agent(src);
    }
    if (!ok) pv = 0.f;
    const float a = wave_sum_dpp(lane < kPR ? pv : 0.f), b = wave_sum_dpp(lane < kPR ? 0.f : pv);
    if (lane == 0) { L.tot[0] = a; L.tot[1] = b; if (!ok) *L.abort = 1; }
  }
}
__device__ __forceinline__ gfloat* word_of(gfloat* pring, int frame, int rank, int word) {
  return pring + (((frame % 3) * kPR + rank) * kPWords + word);
}

// Vector in global memory -> LDS table by LDS-DMA, 1 KB rows dealt to the waves round robin (complete vector; the
// reader's L1 may hold lines of the buffer's previous use).  R is rounded up to whole 16-byte granules: source and table
// are padded accordingly.
__device__ __forceinline__ void dma_table(cgfloat* src, int R, float* table) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  invalidate_l1();
  for (int off = w * 256; off < R; off += kPW * 256)
    if (off + lane * 4 < R) dma256(src + off + lane * 4, table + off);
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


}  // namespace pk2

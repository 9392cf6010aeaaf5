// Tile staging shared by the f32 MFMA GEMM (gemm_f32.hip) and the large-batch LSTM step kernels (lstm.hip):
// operands are staged through LDS in k-major order ([k][row]) whatever their memory layout, so every MFMA
// operand fetch is a conflict-free ds_read_b32 of 32 consecutive floats per half-wave.
#pragma once
#include <type_traits>

#include "common.h"

namespace pk2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Workgroup barrier for LDS hand-overs.  __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`: it also drains
// every global load the thread has in flight -- i.e. the k-slabs prefetched for LATER iterations, which makes every slab
// of the main loop wait a full round trip to memory however deep the prefetch is (measured: ~0.9 us per slab with one
// workgroup per CU, whatever the number of slabs in flight).  The staging loops only exchange data through LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int BK = 16;
constexpr int kGemmThreads = 256;
// TILES = MFMA tiles per wave per dimension: block tile (64*TILES)^2.  TILES = 2 (128x128) for large
// grids; TILES = 1 (64x64) when 128x128 tiles would leave the 256 CUs with fewer than ~2 workgroups
// each (one wave per SIMD cannot hide its own LDS/global latency).
template <int TILES> struct Geo {
  static constexpr int BMN = 64 * TILES;        // block tile edge
  static constexpr int LD = BMN + 4;            // padded leading dimension of the [BK][BMN] LDS tiles (float4-aligned rows)
  // Leading dimension when the operand is k-contiguous in memory: its LDS image is written with scalar stores,
  // lane l -> (k = 4*(l&3)+j, row = l>>2).  LD = 2 (mod 8) spreads the four k's of a 32-lane half over all 32 banks
  // (bank = 8*(l&3) + 2j + row); with LD = 4 (mod 32) lanes l and l+2 collide (2-way conflict on every store).
  static constexpr int LDK = BMN + 2;
  static constexpr int LPK = 16 * TILES;        // lanes covering one k-row when the row dim is contiguous
  template <bool KCONTIG> static constexpr int ld() { return KCONTIG ? LDK : LD; }
};

// Loads the 128 x 16 (rows x k) slab of an operand into registers.
//   KCONTIG: element (r, k) at base[r*ld + k]  -> thread owns float4 along k of 2 rows
//  !KCONTIG: element (r, k) at base[k*ld + r]  -> thread owns float4 along r of 2 k's
// Out-of-range elements read as 0.  `vec` = base/ld allow aligned float4 loads.
template <bool KCONTIG, int TILES>
__device__ __forceinline__ void load_slab(const float* __restrict__ base, int64_t ld, int r0, int k0,
                                          int R, int K, bool vec, float4 (&reg)[TILES]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int h = 0; h < TILES; ++h) {
    int r, k;
    if (KCONTIG) { r = (tid >> 2) + h * 64; k = (tid & 3) * 4; }
    else         { k = tid / Geo<TILES>::LPK + h * (kGemmThreads / Geo<TILES>::LPK); r = (tid % Geo<TILES>::LPK) * 4; }
    const int gr = r0 + r, gk = k0 + k;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KCONTIG) {
      if (gr < R) {
        const float* p = base + (int64_t)gr * ld + gk;
        if (vec && gk + 3 < K) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gk + 0 < K) v.x = p[0];
          if (gk + 1 < K) v.y = p[1];
          if (gk + 2 < K) v.z = p[2];
          if (gk + 3 < K) v.w = p[3];
        }
      }
    } else {
      if (gk < K) {
        const float* p = base + (int64_t)gk * ld + gr;
        if (vec && gr + 3 < R) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gr + 0 < R) v.x = p[0];
          if (gr + 1 < R) v.y = p[1];
          if (gr + 2 < R) v.z = p[2];
          if (gr + 3 < R) v.w = p[3];
        }
      }
    }
    reg[h] = v;
  }
}


// Interior tiles (whole 64*TILES rows inside the matrix, aligned float4 loads, whole k-slab inside K): one pointer per
// float4 computed ONCE, then `ptr + kt * step` per slab -- no bounds tests, no 64-bit multiplies in the main loop (the
// general load_slab costs ~60 VALU / SALU instructions and a dozen exec-mask branches per call; with four calls per slab it
// throttled the MFMA issue: 46 % of peak on the model's shapes against 67-77 % for the same tile shape without it).
//
// Edge tiles (rows of the tile beyond the matrix's R rows) take the same path: a row index beyond the matrix is CLAMPED to
// its last row (to its last aligned group of four rows when the rows are the contiguous dimension), so the loads stay
// aligned and in bounds and the tile multiplies duplicates of valid rows into accumulator rows / columns the epilogue never
// stores.  (Until round 4 edge tiles went through the general loader: on R = 2356 frames the last row band of 64x64 tiles
// ran its main loop 4.6 x as long as the interior ones -- 60 us against 13 in the profile build -- and every product with
// M = frames waited for it: 2356 x 512 x 512 took 33 us of which the interior tiles needed 15.)
template <bool KCONTIG, int TILES>
__device__ __forceinline__ void slab_pointers(const float* __restrict__ base, int64_t ld, int r0, int k0, int R,
                                              const float* (&ptr)[TILES], int64_t* step) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int h = 0; h < TILES; ++h) {
    if (KCONTIG) {
      const int r = (tid >> 2) + h * 64, k = (tid & 3) * 4;
      ptr[h] = base + (int64_t)min(r0 + r, R - 1) * ld + k0 + k;
    } else {
      const int k = tid / Geo<TILES>::LPK + h * (kGemmThreads / Geo<TILES>::LPK), r = (tid % Geo<TILES>::LPK) * 4;
      ptr[h] = base + (int64_t)(k0 + k) * ld + min(r0 + r, R - 4);
    }
  }
  *step = KCONTIG ? (int64_t)BK : (int64_t)BK * ld;
}
template <int TILES>
__device__ __forceinline__ void load_slab_fast(const float* const (&ptr)[TILES], int64_t off, float4 (&reg)[TILES]) {
#pragma unroll
  for (int h = 0; h < TILES; ++h) reg[h] = *reinterpret_cast<const float4*>(ptr[h] + off);
}

template <bool KCONTIG, int TILES>
__device__ __forceinline__ void store_slab(float* __restrict__ tile, const float4 (&reg)[TILES]) {
  const int tid = threadIdx.x;
  constexpr int LD = Geo<TILES>::template ld<KCONTIG>();
#pragma unroll
  for (int h = 0; h < TILES; ++h) {
    if (KCONTIG) {
      const int r = (tid >> 2) + h * 64, k = (tid & 3) * 4;
      tile[(k + 0) * LD + r] = reg[h].x;
      tile[(k + 1) * LD + r] = reg[h].y;
      tile[(k + 2) * LD + r] = reg[h].z;
      tile[(k + 3) * LD + r] = reg[h].w;
    } else {
      const int k = tid / Geo<TILES>::LPK + h * (kGemmThreads / Geo<TILES>::LPK), r = (tid % Geo<TILES>::LPK) * 4;
      *reinterpret_cast<float4*>(&tile[k * LD + r]) = reg[h];
    }
  }
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}

// The pipelined main loop of a block tile: acc (zeroed here) = A[m0.., kbeg..K) * B[kbeg..K), n0..] for the
// (64 TILES) x (64 TILES) tile of 256 threads (4 waves, 2 x 2; wave w owns rows 32 TILES (w / 2).., columns 32 TILES (w % 2)..).
// KCA / KCB: the operand is k-contiguous in memory (element (r, k) at base[r ld + k]) or row-contiguous (base[k ld + r]).
// As / Bs: two LDS buffers of BK x Geo<TILES>::ld<KC?>() floats each.  Ends behind a barrier (the buffers are free).
template <bool KCA, bool KCB, int TILES>
__device__ __forceinline__ void tile_mainloop(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                              int m0, int n0, int kbeg, int K, int M, int N, bool vecA, bool vecB,
                                              float (*As)[BK * Geo<TILES>::template ld<KCA>()],
                                              float (*Bs)[BK * Geo<TILES>::template ld<KCB>()],
                                              f32x16 (&acc)[TILES][TILES], bool zero_acc = true) {
  constexpr int BM = Geo<TILES>::BMN, BN = Geo<TILES>::BMN;
  constexpr int LDA = Geo<TILES>::template ld<KCA>(), LDB = Geo<TILES>::template ld<KCB>();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = (w >> 1) * 32 * TILES, wn = (w & 1) * 32 * TILES;
  if (zero_acc) {        // (false: a further segment of pk2_gemm_f32_seg adds onto the accumulators of the one before)
#pragma unroll
    for (int i = 0; i < TILES; ++i)
#pragma unroll
      for (int j = 0; j < TILES; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  const int kq = lane >> 5, li = lane & 31;
  // DEPTH k-slabs of BK = 16 in flight in registers, two LDS buffers, ONE barrier per slab: while slab kt is multiplied
  // out of LDS[kt & 1], slab kt+1 (loaded DEPTH iterations ago) is written into the other buffer and slabs kt+2 ..
  // kt+DEPTH+1 are on their way from memory.  (Round 1: one LDS buffer, one slab in flight, two barriers per slab -- the
  // global-load latency of a slab was exposed at every barrier: 38 % MFMA-busy on the model's shapes.)  A 128x128 tile
  // multiplies for ~0.85 us per slab: two slabs in flight cover a round trip to memory.  A 64x64 tile multiplies for
  // ~0.2 us, and products small enough to get 64x64 tiles put one or two workgroups on a CU: six slabs in flight.
#ifndef PK2_GEMM_DEPTH2
#define PK2_GEMM_DEPTH2 2
#endif
#ifndef PK2_GEMM_DEPTH1
#define PK2_GEMM_DEPTH1 6
#endif
  constexpr int DEPTH = TILES == 1 ? PK2_GEMM_DEPTH1 : PK2_GEMM_DEPTH2;
  static_assert(DEPTH % 2 == 0, "an even number of register stages (the LDS buffer is the stage's parity)");
  float4 ra[DEPTH][TILES], rb[DEPTH][TILES];
  const int nk = (K - kbeg + BK - 1) / BK;
  // Interior tile (whole rows inside the matrix, aligned float4 loads): the slabs that lie wholly inside K are loaded
  // through precomputed pointers (gemm_tile.h) by straight-line code -- with a branch inside the fetch the compiler can no
  // longer count the loads in flight and waits for all of them (vmcnt(0)) before every LDS store, which exposes a round
  // trip to memory per slab whatever DEPTH is; a last partial slab is multiplied separately behind the loop.
  // (an edge tile qualifies when its rows can be clamped: always for a k-contiguous operand, for a row-contiguous one when
  // the row count is a multiple of four -- slab_pointers)
  const bool interior = vecA && vecB && (m0 + BM <= M || KCA || ((M & 3) == 0 && M >= 4)) &&
                        (n0 + BN <= N || KCB || ((N & 3) == 0 && N >= 4));
  const float* pa[TILES]; const float* pb[TILES];
  int64_t stepA = 0, stepB = 0;
  slab_pointers<KCA, TILES>(A, lda, m0, kbeg, M, pa, &stepA);
  slab_pointers<KCB, TILES>(B, ldb, n0, kbeg, N, pb, &stepB);
#ifndef PK2_GEMM_HOIST
#define PK2_GEMM_HOIST 8        // k-pairs whose operand reads are issued together ahead of their MFMAs (64x64 tiles)
#endif
  auto multiply = [&](auto C_) {
    constexpr int cur = decltype(C_)::value;
    // GRP k-pairs at a time: all their operand reads, then their MFMAs.  (Written as "read a pair, multiply it" the
    // compiler reuses the same registers for every pair and waits lgkmcnt(0) before every two MFMAs of a 64x64 tile: an LDS
    // round trip of ~100 clocks exposed per 128 clocks of MFMA time.  The 128x128 tiles have eight MFMAs per pair and keep
    // the plain order: the extra operand registers cost them more than the exposed round trip.)
    constexpr int GRP = TILES == 1 ? PK2_GEMM_HOIST : 1;
#pragma unroll
    for (int k0 = 0; k0 < BK / 2; k0 += GRP) {
      float a[GRP][TILES], b[GRP][TILES];
#pragma unroll
      for (int g = 0; g < GRP; ++g) {
#pragma unroll
        for (int i = 0; i < TILES; ++i) a[g][i] = As[cur][(2 * (k0 + g) + kq) * LDA + wm + i * 32 + li];
#pragma unroll
        for (int j = 0; j < TILES; ++j) b[g][j] = Bs[cur][(2 * (k0 + g) + kq) * LDB + wn + j * 32 + li];
      }
      if (GRP > 1) __builtin_amdgcn_sched_barrier(0);      // (the scheduler moves the reads back down to their MFMAs otherwise)
#pragma unroll
      for (int g = 0; g < GRP; ++g)
#pragma unroll
        for (int i = 0; i < TILES; ++i)
#pragma unroll
          for (int j = 0; j < TILES; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][i], b[g][j], acc[i][j], 0, 0, 0);
    }
  };
  // the pipelined loop over `n` slabs; slab i travels in register stage i % DEPTH
  auto pipeline = [&](int n, auto fetch) {
    if (n <= 0) return;
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) fetch(min(i, n - 1), ra[i], rb[i]);
    store_slab<KCA, TILES>(As[0], ra[0]);
    store_slab<KCB, TILES>(Bs[0], rb[0]);
    lds_barrier();
    fetch(min(DEPTH, n - 1), ra[0], rb[0]);
    auto slab_step = [&](int kt, auto P) {
      constexpr int st = decltype(P)::value, cur = st & 1, nxt = 1 - cur, sn = (st + 1) % DEPTH;
      multiply(std::integral_constant<int, cur>());
      if (kt + 1 < n) {
        store_slab<KCA, TILES>(As[nxt], ra[sn]);
        store_slab<KCB, TILES>(Bs[nxt], rb[sn]);
      }
      lds_barrier();        // (not __syncthreads: the slabs in flight must stay in flight)
      if (kt + 1 + DEPTH < n) fetch(kt + 1 + DEPTH, ra[sn], rb[sn]);
    };
    // Whole groups of DEPTH steps run without a branch between the loads and the waits for them, so the compiler waits
    // for exactly the slab it is about to store (vmcnt(in flight behind it)) instead of for everything: behind the last
    // slab the fetch index is clamped (the last slab is fetched again: a few cache hits) and the store of a slab that
    // does not exist goes into the LDS buffer nobody reads any more.
    auto steady_step = [&](int kt, auto P) {
      constexpr int st = decltype(P)::value, cur = st & 1, nxt = 1 - cur, sn = (st + 1) % DEPTH;
      multiply(std::integral_constant<int, cur>());
      store_slab<KCA, TILES>(As[nxt], ra[sn]);
      store_slab<KCB, TILES>(Bs[nxt], rb[sn]);
      lds_barrier();
      fetch(min(kt + 1 + DEPTH, n - 1), ra[sn], rb[sn]);
    };
    int kt = 0;
    for (; kt + DEPTH <= n; kt += DEPTH)
      static_for<0, DEPTH>([&](auto S) { steady_step(kt + decltype(S)::value, S); });
    if (kt < n)                         // a last, partial group
      static_for<0, DEPTH>([&](auto S) { if (kt + decltype(S)::value < n) slab_step(kt + decltype(S)::value, S); });
  };
  if (interior) {
    const int nk_fast = (K - kbeg) / BK;        // slabs that lie wholly inside K
    pipeline(nk_fast, [&](int ks, float4 (&xa)[TILES], float4 (&xb)[TILES]) {
      load_slab_fast<TILES>(pa, ks * stepA, xa);
      load_slab_fast<TILES>(pb, ks * stepB, xb);
    });
    if (nk > nk_fast) {     // the partial last slab (its LDS buffers are free: the loop ends behind a barrier)
      load_slab<KCA, TILES>(A, lda, m0, kbeg + nk_fast * BK, M, K, vecA, ra[0]);
      load_slab<KCB, TILES>(B, ldb, n0, kbeg + nk_fast * BK, N, K, vecB, rb[0]);
      store_slab<KCA, TILES>(As[0], ra[0]);
      store_slab<KCB, TILES>(Bs[0], rb[0]);
      lds_barrier();
      multiply(std::integral_constant<int, 0>());
      lds_barrier();        // (stream-K: the next piece of this workgroup reuses the buffers)
    }
  } else {
    pipeline(nk, [&](int ks, float4 (&xa)[TILES], float4 (&xb)[TILES]) {
      load_slab<KCA, TILES>(A, lda, m0, kbeg + ks * BK, M, K, vecA, xa);
      load_slab<KCB, TILES>(B, ldb, n0, kbeg + ks * BK, N, K, vecB, xb);
    });
  }
}

}  // namespace pk2

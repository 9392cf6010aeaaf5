// Tile staging shared by the f32 MFMA GEMM (gemm_f32.hip) and the large-batch LSTM step kernels (lstm.hip):
// operands are staged through LDS in k-major order ([k][row]) whatever their memory layout, so every MFMA
// operand fetch is a conflict-free ds_read_b32 of 32 consecutive floats per half-wave.
#pragma once
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
template <bool KCONTIG, int TILES>
__device__ __forceinline__ void slab_pointers(const float* __restrict__ base, int64_t ld, int r0, int k0,
                                              const float* (&ptr)[TILES], int64_t* step) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int h = 0; h < TILES; ++h) {
    if (KCONTIG) {
      const int r = (tid >> 2) + h * 64, k = (tid & 3) * 4;
      ptr[h] = base + (int64_t)(r0 + r) * ld + k0 + k;
    } else {
      const int k = tid / Geo<TILES>::LPK + h * (kGemmThreads / Geo<TILES>::LPK), r = (tid % Geo<TILES>::LPK) * 4;
      ptr[h] = base + (int64_t)(k0 + k) * ld + r0 + r;
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


}  // namespace pk2

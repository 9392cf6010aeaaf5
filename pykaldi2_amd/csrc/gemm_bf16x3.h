// f32-accurate product on the bf16 MFMA of gfx950 (round 6): every f32 operand element is cut ONCE per staged slab into
// three bf16 numbers  x = hi + mid + lo  (hi = bf16_rne(x), mid = bf16_rne(x - hi), lo = bf16_rne(x - hi - mid); the
// differences are exact in f32 and 3 x 8 significant bits cover the 24 of an f32, so the sum is exact), and a k-step
// multiplies six of the nine part products on v_mfma_f32_32x32x16_bf16 into ONE f32 accumulator:
//      hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi          (dropped: mid*lo, lo*mid, lo*lo, each <= 2^-24 |a b|).
// Products of two bf16 numbers are exact in f32; what remains is the f32 accumulation -- one rounding per MFMA (16 k's)
// instead of one per k as on v_mfma_f32_32x32x2_f32.  tools/bf16x3_error_study.py (CPU model, the model's shapes and
// value distributions): max and rms error against float64 are 0.4-0.5 x the f32 product's; dropped terms alone 1e-9 of
// sum |a b|.  Six bf16 MFMAs of 32 cycles replace eight f32 MFMAs of 64 cycles per 16 k's: 2.67 x the f32 MFMA rate
// (2.5 PFLOP/s / 6 = 417 TFLOP/s of f32-equivalent work against 157).
//
// Same outer shape as gemm_tile.h (reference: the cuBLAS SGEMMs under nn.Linear / nn.LSTM, models/lstm.py:45-59):
// (64 TILES)^2 block tile, 4 waves 2 x 2, two LDS buffers, DEPTH register stages, one LDS-only barrier per slab.  What
// differs is the LDS image: per operand three planes (hi, mid, lo) of 16-byte slots  [plane][k-group of 8][row slot],
// a slot = the 8 bf16 of one row for 8 consecutive k's = exactly one lane's MFMA operand (lanes 0-31: k-group 2s,
// lanes 32-63: k-group 2s+1 of MFMA step s), read with one ds_read_b128.
//   * k-contiguous operand (element (r, k) at base[r ld + k]): a thread owns float4s along k; 4 bf16 = 8 bytes per
//     plane, ds_write_b64 into slot(row) = row.  Row pitch of a k-group = BMN + 4 slots: the 16 lanes of a store group
//     (4 rows x 4 k-quads) land on 16 distinct 8-byte bank pairs.
//   * row-contiguous operand (base[k ld + r]): a thread owns the float4s of TWO consecutive k's over the same 4 rows
//     and packs (k, k+1) pairs: one dword per plane and row, ds_write_b32.  The transposition happens here, in the
//     address: slot(row) = row/4 + SW (row%4) with SW = BMN/4 + 4, and the thread mapping puts 8 row-quads x 4 k-pairs
//     into a 32-lane store group, so the 32 dwords fall on 32 distinct banks; the ds_read_b128 of 32 consecutive rows
//     then touches slot (row/4 + 4 (row%4)) mod 16 -- all 16 distinct inside each of the instruction's four 16-lane
//     service groups ({0-3,12-15,20-27}, ...), i.e. conflict-free on both sides without a transpose read.
// Inf / NaN: an operand element that is +-inf (or rounds to inf in bf16: |x| > 3.39e38) yields NaN (inf - inf in the
// split) where the f32 product would give inf; finite data, which is all the models produce, is unaffected.
#pragma once
#include "gemm_tile.h"

namespace pk2 {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <int TILES> struct GeoX {
  static constexpr int BMN = 64 * TILES;
  static constexpr int BK = TILES == 2 ? 16 : 32;     // every thread stages exactly two float4 per operand and slab
  static constexpr int NKG = BK / 8;                  // k-groups (16-byte slots along k) per slab
  static constexpr int NQ = BMN / 4;                  // row quads
  static constexpr int SW = NQ + 4;                   // slot stride of (row % 4) in the swizzled (row-contiguous) image
  template <bool KC> static constexpr int sp() { return KC ? BMN + 4 : NQ + 3 * SW; }     // slots per k-group
  template <bool KC> static constexpr int slots() { return 3 * NKG * sp<KC>(); }          // 16-byte slots per operand stage
  template <bool KC> __device__ static __forceinline__ int slot(int row) { return KC ? row : (row >> 2) + SW * (row & 3); }
};

// (a, b) -> one dword of two bf16, round-to-nearest-even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t v = {a, b};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(unsigned, r);
}
// The three-way split of a pair: dwords of (hi, hi), (mid, mid), (lo, lo).
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = pk_bf16(x0, x1);
#ifdef PK2_X3_DBG_NOSPLIT       // (bottleneck hunt: wrong results, no split arithmetic)
  m = h; l = h; return;
#endif
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = pk_bf16(s0, s1);
}

// Thread -> piece maps.  k-contiguous: piece h = float4 at (row, 4 kquad..).  Row-contiguous: piece h = float4 at
// (k = 8 kg + 2 d + h, rows 4 q..4 q + 3).
template <bool KC, int TILES> struct PieceMap {
  int row[2], k[2];     // k-contiguous: both used.  Row-contiguous: row[0] = 4 q, k[0] = 8 kg + 2 d (k[1] = k[0] + 1)
  int kg, d;
  __device__ __forceinline__ PieceMap() {
    const int t = threadIdx.x;
    if (KC) {
      if (TILES == 2) { row[0] = t >> 2; row[1] = (t >> 2) + 64; k[0] = k[1] = (t & 3) * 4; }
      else            { row[0] = row[1] = t >> 2; k[0] = (t & 3) * 4; k[1] = k[0] + 16; }
      kg = d = 0;
    } else {
      const int q = (t & 7) + 8 * ((t >> 5) & (TILES == 2 ? 3 : 1));
      d = (t >> 3) & 3;
      kg = TILES == 2 ? (t >> 7) : (t >> 6);
      row[0] = row[1] = 4 * q;
      k[0] = 8 * kg + 2 * d;
      k[1] = k[0] + 1;
    }
  }
};

template <bool KC, int TILES>
__device__ __forceinline__ void xslab_pointers(const PieceMap<KC, TILES>& pm, const float* __restrict__ base, int64_t ld, int r0,
                                               int k0, int R, const float* (&ptr)[2], int64_t* step) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (KC) ptr[h] = base + (int64_t)min(r0 + pm.row[h], R - 1) * ld + k0 + pm.k[h];
    else    ptr[h] = base + (int64_t)(k0 + pm.k[h]) * ld + min(r0 + pm.row[h], R - 4);
  }
  *step = KC ? (int64_t)GeoX<TILES>::BK : (int64_t)GeoX<TILES>::BK * ld;
}

// General (bounds-checked, zero-filling) loader of a thread's two pieces.
template <bool KC, int TILES>
__device__ __forceinline__ void xload_general(const PieceMap<KC, TILES>& pm, const float* __restrict__ base, int64_t ld, int r0,
                                              int k0, int R, int K, bool vec, float4 (&reg)[2]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int gr = r0 + pm.row[h], gk = k0 + pm.k[h];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {
      if (gr < R) {
        const float* p = base + (int64_t)gr * ld + gk;
        if (vec && gk + 3 < K) v = *reinterpret_cast<const float4*>(p);
        else {
          if (gk + 0 < K) v.x = p[0];
          if (gk + 1 < K) v.y = p[1];
          if (gk + 2 < K) v.z = p[2];
          if (gk + 3 < K) v.w = p[3];
        }
      }
    } else {
      if (gk < K) {
        const float* p = base + (int64_t)gk * ld + gr;
        if (vec && gr + 3 < R) v = *reinterpret_cast<const float4*>(p);
        else {
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

// Split a thread's two float4 and write the three planes of the stage `tile` (16-byte slots).
// CS (row-contiguous A of a weight-gradient product only): cs[i] += the two values of row 4 q + i -- the thread's share of the
// operand's sums over k, i.e. of the bias gradient that goes with the weight gradient (see tile_mainloop_bf16x3).
template <bool KC, int TILES, bool CS = false>
__device__ __forceinline__ void xstore(const PieceMap<KC, TILES>& pm, u32x4_t* __restrict__ tile, const float4 (&reg)[2],
                                       float* cs = nullptr, float csw = 0.f) {
  typedef GeoX<TILES> G;
  constexpr int SP = G::template sp<KC>(), PL = G::NKG * SP;      // slots per k-group / per plane
  if (KC) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      unsigned h0, m0, l0, h1, m1, l1;
      split_pair(reg[h].x, reg[h].y, h0, m0, l0);
      split_pair(reg[h].z, reg[h].w, h1, m1, l1);
      const int kq = pm.k[h] >> 2;                                 // k-quad inside the slab
      uint2* p = reinterpret_cast<uint2*>(tile + (kq >> 1) * SP + pm.row[h]) + (kq & 1);
      p[0] = make_uint2(h0, h1);
      p[2 * PL] = make_uint2(m0, m1);
      p[4 * PL] = make_uint2(l0, l1);
    }
  } else {
    const float a[4] = {reg[0].x, reg[0].y, reg[0].z, reg[0].w};
    const float b[4] = {reg[1].x, reg[1].y, reg[1].z, reg[1].w};
    unsigned* p = reinterpret_cast<unsigned*>(tile + pm.kg * SP + (pm.row[0] >> 2)) + pm.d;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned hh, mm, ll;
      split_pair(a[i], b[i], hh, mm, ll);
      p[4 * (G::SW * i)] = hh;
      p[4 * (G::SW * i + PL)] = mm;
      p[4 * (G::SW * i + 2 * PL)] = ll;
      if (CS) cs[i] = fmaf(csw, a[i] + b[i], cs[i]);      // (csw = 0: the pipeline's clamped re-store of the last slab)
    }
  }
}

// acc (zeroed here) = A[m0.., kbeg..K) * B[kbeg..K), n0..], f32-accurate through the three-way bf16 split.
// As / Bs: two LDS stages of GeoX<TILES>::slots<KC?>() 16-byte slots each.  Ends behind a barrier.
template <bool KCA, bool KCB, int TILES>
__device__ __forceinline__ void tile_mainloop_bf16x3(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                     int m0, int n0, int kbeg, int K, int M, int N, bool vecA, bool vecB,
                                                     u32x4_t (*As)[GeoX<TILES>::template slots<KCA>()],
                                                     u32x4_t (*Bs)[GeoX<TILES>::template slots<KCB>()],
                                                     f32x16 (&acc)[TILES][TILES], float* __restrict__ colsum = nullptr,
                                                     bool zero_acc = true) {
  // colsum (row-contiguous A only, i.e. the weight-gradient form dW = dY^T X): colsum[m] += sum_k A[k][m] over this tile's k
  // range, added with float atomics -- the bias gradient rides in the product that reads dY anyway (round 6: 62 colsum
  // launches per TransformerAM step, one per LF-MMI step).  The caller passes it to the tiles of ONE column block only.
  typedef GeoX<TILES> G;
  constexpr int BM = G::BMN, BN = G::BMN, XBK = G::BK, NKG = G::NKG;
  constexpr int SPA = G::template sp<KCA>(), SPB = G::template sp<KCB>();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = (w >> 1) * 32 * TILES, wn = (w & 1) * 32 * TILES;
  if (zero_acc) {
#pragma unroll
    for (int i = 0; i < TILES; ++i)
#pragma unroll
      for (int j = 0; j < TILES; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  const int kq = lane >> 5, li = lane & 31;
#ifndef PK2_GEMMX_DEPTH
#define PK2_GEMMX_DEPTH 2
#endif
  constexpr int DEPTH = PK2_GEMMX_DEPTH;
  static_assert(DEPTH % 2 == 0, "an even number of register stages (the LDS buffer is the stage's parity)");
  float4 ra[DEPTH][2], rb[DEPTH][2];
  const PieceMap<KCA, TILES> pma;
  const PieceMap<KCB, TILES> pmb;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_cs = !KCA && colsum != nullptr;
  auto store_a = [&](u32x4_t* tile, const float4 (&reg)[2], bool real = true) {
    if constexpr (!KCA) { if (do_cs) { xstore<KCA, TILES, true>(pma, tile, reg, cs, real ? 1.f : 0.f); return; } }
    xstore<KCA, TILES>(pma, tile, reg);
  };
  const int nk = (K - kbeg + XBK - 1) / XBK;
  const bool interior = vecA && vecB && (m0 + BM <= M || KCA || ((M & 3) == 0 && M >= 4)) &&
                        (n0 + BN <= N || KCB || ((N & 3) == 0 && N >= 4));
  const float* pa[2]; const float* pb[2];
  int64_t stepA = 0, stepB = 0;
  xslab_pointers<KCA, TILES>(pma, A, lda, m0, kbeg, M, pa, &stepA);
  xslab_pointers<KCB, TILES>(pmb, B, ldb, n0, kbeg, N, pb, &stepB);
  // per-lane operand slots of this wave's MFMA tiles (k-group kq of step 0)
  int sa[TILES], sb[TILES];
#pragma unroll
  for (int i = 0; i < TILES; ++i) sa[i] = kq * SPA + G::template slot<KCA>(wm + i * 32 + li);
#pragma unroll
  for (int j = 0; j < TILES; ++j) sb[j] = kq * SPB + G::template slot<KCB>(wn + j * 32 + li);
  auto multiply = [&](auto C_) {
    constexpr int cur = decltype(C_)::value;
#pragma unroll
    for (int s = 0; s < NKG / 2; ++s) {
      bf16x8_t a[3][TILES], b[3][TILES];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TILES; ++i) a[p][i] = __builtin_bit_cast(bf16x8_t, As[cur][(p * NKG + 2 * s) * SPA + sa[i]]);
#pragma unroll
        for (int j = 0; j < TILES; ++j) b[p][j] = __builtin_bit_cast(bf16x8_t, Bs[cur][(p * NKG + 2 * s) * SPB + sb[j]]);
      }
      // small terms first; consecutive MFMAs go to different accumulators
#define PK2_X3(PA, PB)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < TILES; ++i) _Pragma("unroll") for (int j = 0; j < TILES; ++j)    \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][i], b[PB][j], acc[i][j], 0, 0, 0)
#ifdef PK2_X3_DBG_ONEMFMA      // (bottleneck hunt: one product instead of six)
      PK2_X3(0, 0);
      asm volatile("" :: "v"(a[1][0]), "v"(a[2][0]), "v"(b[1][0]), "v"(b[2][0]));
#else
      PK2_X3(2, 0); PK2_X3(0, 2); PK2_X3(1, 1); PK2_X3(1, 0); PK2_X3(0, 1); PK2_X3(0, 0);
#endif
#undef PK2_X3
    }
  };
  auto pipeline = [&](int n, auto fetch) {
    if (n <= 0) return;
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) fetch(min(i, n - 1), ra[i], rb[i]);
    store_a(As[0], ra[0]);
    xstore<KCB, TILES>(pmb, Bs[0], rb[0]);
    lds_barrier();
    fetch(min(DEPTH, n - 1), ra[0], rb[0]);
    // (the scheduler otherwise sinks a fetch into the middle of the NEXT step's MFMAs to shorten the live range of its
    // registers: the loads then have half a step -- ~0.15 us -- to come back instead of DEPTH - 1/2 steps)
#ifndef PK2_GEMMX_PIN
#define PK2_GEMMX_PIN 1
#endif
    auto pin = [] { if (PK2_GEMMX_PIN) __builtin_amdgcn_sched_barrier(0); };
    auto slab_step = [&](int kt, auto P) {
      constexpr int st = decltype(P)::value, cur = st & 1, nxt = 1 - cur, sn = (st + 1) % DEPTH;
      multiply(std::integral_constant<int, cur>());
      if (kt + 1 < n) {
        store_a(As[nxt], ra[sn]);
        xstore<KCB, TILES>(pmb, Bs[nxt], rb[sn]);
      }
      lds_barrier();
      if (kt + 1 + DEPTH < n) fetch(kt + 1 + DEPTH, ra[sn], rb[sn]);
    };
    auto steady_step = [&](int kt, auto P) {
      constexpr int st = decltype(P)::value, cur = st & 1, nxt = 1 - cur, sn = (st + 1) % DEPTH;
      multiply(std::integral_constant<int, cur>());
      store_a(As[nxt], ra[sn], kt + 1 < n);
      xstore<KCB, TILES>(pmb, Bs[nxt], rb[sn]);
      lds_barrier();
      fetch(min(kt + 1 + DEPTH, n - 1), ra[sn], rb[sn]);
      pin();
    };
    int kt = 0;
    for (; kt + DEPTH <= n; kt += DEPTH)
      static_for<0, DEPTH>([&](auto S) { steady_step(kt + decltype(S)::value, S); });
    if (kt < n)
      static_for<0, DEPTH>([&](auto S) { if (kt + decltype(S)::value < n) slab_step(kt + decltype(S)::value, S); });
  };
  if (interior) {
    const int nk_fast = (K - kbeg) / XBK;
    pipeline(nk_fast, [&](int ks, float4 (&xa)[2], float4 (&xb)[2]) {
#ifdef PK2_X3_DBG_NOLOAD        // (bottleneck hunt: every slab re-reads slab 0 -- L1 hits)
      ks = 0;
#endif
#pragma unroll
      for (int h = 0; h < 2; ++h) xa[h] = *reinterpret_cast<const float4*>(pa[h] + ks * stepA);
#pragma unroll
      for (int h = 0; h < 2; ++h) xb[h] = *reinterpret_cast<const float4*>(pb[h] + ks * stepB);
    });
    if (nk > nk_fast) {
      xload_general<KCA, TILES>(pma, A, lda, m0, kbeg + nk_fast * XBK, M, K, vecA, ra[0]);
      xload_general<KCB, TILES>(pmb, B, ldb, n0, kbeg + nk_fast * XBK, N, K, vecB, rb[0]);
      store_a(As[0], ra[0]);
      xstore<KCB, TILES>(pmb, Bs[0], rb[0]);
      lds_barrier();
      multiply(std::integral_constant<int, 0>());
      lds_barrier();
    }
  } else {
    pipeline(nk, [&](int ks, float4 (&xa)[2], float4 (&xb)[2]) {
      xload_general<KCA, TILES>(pma, A, lda, m0, kbeg + ks * XBK, M, K, vecA, xa);
      xload_general<KCB, TILES>(pmb, B, ldb, n0, kbeg + ks * XBK, N, K, vecB, xb);
    });
  }
  if constexpr (!KCA) {
    if (do_cs) {      // the threads that share a row quad (k-groups x k-pairs) add up in LDS (the stages are free: the loop ends
      float* red = reinterpret_cast<float*>(&As[0][0]);     // behind a barrier), then one atomic per row of the matrix
      for (int r = threadIdx.x; r < BM; r += kGemmThreads) red[r] = 0.f;
      lds_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) atomicAdd(&red[pma.row[0] + i], cs[i]);
      lds_barrier();
      for (int r = threadIdx.x; r < BM; r += kGemmThreads)
        if (m0 + r < M) atomicAdd(colsum + m0 + r, red[r]);
      lds_barrier();
    }
  }
}


// Round 6, measured and removed: a wide block tile (256 x 128, 512 threads = 8 waves 4 x 2, one workgroup per CU; a quarter
// less operand traffic, split arithmetic and LDS stores per MFMA).  Parity-green in all four layouts and no faster on any
// product of the CE configuration (20480 x 4096 x 1024: 954 us against 925).  The counters say why (tools/gpu_x3_pmc2.sh,
// profiles/r06_gemm_bf16x3_pmc.txt): on random operands BOTH kernels run against the chip's power limit -- the 128 x 128 kernel
// needs 1.59 M cycles per CU at 1.52 GHz (profiled run), the wide one 1.99 M cycles at 1.89 GHz, the same 1.05 ms; on ZERO
// operands (nothing toggles, 2.06-2.12 GHz) 805 against 988 us.  What bounds the bf16x3 product is the energy of its six
// MFMAs per k-step on random mantissas: 185 TFLOP/s of f32-equivalent work = 1.11 PFLOP/s of executed bf16 work, where
// tuned plain-bf16 GEMMs on this part reach ~1.25 PFLOP/s on random data (MI355X_MICROARCH.md, DVFS give-back).

}  // namespace pk2

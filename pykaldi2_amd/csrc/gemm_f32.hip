// f32 MFMA GEMM for gfx950:  C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C + bias[N].
//
// Replaces the cuBLAS GEMMs under nn.Linear / nn.LSTM's input projections (reference
// models/lstm.py:45-59).  FP32 inputs and accumulation on v_mfma_f32_32x32x2_f32: the north
// star's 1e-4 posterior tolerance rules out bf16 GEMMs, and gfx950 has
// no TF32-like mode, so the f32 MFMA (157 TFLOP/s peak) is the matrix-core path for this model.
//
// 128x128x16 block tile, 256 threads = 4 wavefronts in a 2x2 grid, each wave owns a 64x64
// sub-tile = 2x2 MFMA tiles (64 accumulator VGPRs).  Operands are staged through LDS in
// k-major order ([k][m] / [k][n]) whatever their memory layout, so every MFMA operand fetch is a
// conflict-free ds_read_b32 of 32 consecutive floats per half-wave.  The next k-tile is
// prefetched into registers while the current one is multiplied.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_tile.h"
#include "gemm_bf16x3.h"

namespace pk2 {

// Batched form: blockIdx.z = i0 * n1 + i1 selects a matrix triple at offsets i0*s?0 + i1*s?1 (floats).
// Split-K form (ksplit > 1): blockIdx.z = batch * ksplit + slice; a slice covers klen k's and adds alpha * its
// partial product into C with float atomics (C already holds beta * C + bias, see gemm_prescale_kernel).
struct GemmBatch { int n1; int64_t sA0, sA1, sB0, sB1, sC0, sC1; int ksplit, klen; float* colsum = nullptr;      // colsum: pk2_gemm_f32_tn_colsum
                   int act = 0; const float* gate = nullptr; int64_t ldg = 0;        // pk2_gemm_f32_act: 1 = ReLU, 2 = kept where gate > 0
                   int nseg = 1; int64_t segA = 0, segB = 0; };                       // pk2_gemm_f32_seg: sum over segments of A / B

// One block tile: C[m0.., n0..] (+)= alpha * A[m0.., kbeg..kend) * B[kbeg..kend), n0..].  `atomic`: the tile's k range is
// shared with other workgroups -- the product is added with float atomics into a C that already holds beta * C + bias.
#ifndef PK2_GEMMX_WAVES
#define PK2_GEMMX_WAVES 2       // waves per SIMD the bf16x3 kernels are compiled for
#endif
// Epilogue of a block tile: the wave's TILES x TILES accumulators (C/D layout of the 32x32 MFMA: col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) -> C rows m0 + wm.., columns n0 + wn..
template <int TILES>
__device__ __forceinline__ void gemm_store(const f32x16 (&acc)[TILES][TILES], int M, int N, float alpha, float beta, float* __restrict__ C,
                                           int64_t ldc, const float* __restrict__ bias, int m0, int n0, int wm, int wn, bool atomic,
                                           int act = 0, const float* __restrict__ gate = nullptr, int64_t ldg = 0) {
  const int lane = threadIdx.x & 63;
  const int col_l = lane & 31, row_h = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < TILES; ++i) {
#pragma unroll
    for (int j = 0; j < TILES; ++j) {
      const int gc = n0 + wn + j * 32 + col_l;
      if (gc >= N) continue;
      if (atomic) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gr = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
          if (gr < M) atomicAdd(C + (int64_t)gr * ldc + gc, alpha * acc[i][j][r]);
        }
        continue;
      }
      const float bv = bias ? bias[gc] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gr = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
        if (gr < M) {
          float* o = C + (int64_t)gr * ldc + gc;
          float v = alpha * acc[i][j][r] + bv;
          if (beta != 0.f) v += beta * (*o);
          // (pk2_gemm_f32_act: the ReLU behind a Linear, or the ReLU's backward mask from the forward activation)
          if (act == 1) v = fmaxf(v, 0.f);
          else if (act == 2) v = gate[(int64_t)gr * ldg + gc] > 0.f ? v : 0.f;
          *o = v;
        }
      }
    }
  }
}

// LDS bytes of one block tile (two stages of both operands).
template <bool TA, bool TB, int TILES, bool X3>
constexpr size_t gemm_smem_bytes() {
  if (X3) return 2 * 16 * (size_t)(GeoX<TILES>::template slots<!TA>() + GeoX<TILES>::template slots<TB>());
  return 2 * sizeof(float) * BK * (size_t)(Geo<TILES>::template ld<!TA>() + Geo<TILES>::template ld<TB>());
}

template <bool TA, bool TB, int TILES, bool X3, bool SEG = false>
__device__ __forceinline__ void gemm_block(void* smem, int M, int N, int K, int kbeg, float alpha, const float* __restrict__ A, int64_t lda,
                                           const float* __restrict__ B, int64_t ldb, float beta, float* __restrict__ C,
                                           int64_t ldc, const float* __restrict__ bias, bool vecA, bool vecB, int m0, int n0,
                                           bool atomic, float* colsum = nullptr, int act = 0, const float* __restrict__ gate = nullptr,
                                           int64_t ldg = 0, int nseg = 1, int64_t segA = 0, int64_t segB = 0) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 32 * TILES, wn = (w & 1) * 32 * TILES;

  f32x16 acc[TILES][TILES];
#ifdef PK2_GEMM_PROFILE
  const long long gp_t0 = wall_clock64();
#endif
  if constexpr (X3) {       // three-way bf16 split, six products per k-step on the bf16 MFMA (gemm_bf16x3.h)
    typedef u32x4_t (*StA)[GeoX<TILES>::template slots<!TA>()];
    typedef u32x4_t (*StB)[GeoX<TILES>::template slots<TB>()];
    StA Ax = reinterpret_cast<StA>(smem);
    StB Bx = reinterpret_cast<StB>(reinterpret_cast<u32x4_t*>(smem) + 2 * GeoX<TILES>::template slots<!TA>());
    // SEG (pk2_gemm_f32_seg): the segments' products one behind the other into the same accumulators (each main loop ends behind a
    // barrier).  A template parameter, not a run-time branch: with both forms in ONE kernel every ordinary product ran slower (CE
    // step 19.0 -> 19.8 ms, LF-MMI 11.59 -> 11.73, same job twice: the second copy of the main loop changes the register allocation).
    if constexpr (!SEG) tile_mainloop_bf16x3<!TA, TB, TILES>(A, lda, B, ldb, m0, n0, kbeg, K, M, N, vecA, vecB, Ax, Bx, acc, n0 == 0 ? colsum : nullptr);
    else
      for (int sg = 0; sg < nseg; ++sg)
        tile_mainloop_bf16x3<!TA, TB, TILES>(A + sg * segA, lda, B + sg * segB, ldb, m0, n0, kbeg, K, M, N, vecA, vecB, Ax, Bx, acc, nullptr, sg == 0);
  } else {
    constexpr int LDA = Geo<TILES>::template ld<!TA>(), LDB = Geo<TILES>::template ld<TB>();   // per-operand LDS row pitch
    typedef float (*StA)[BK * LDA];
    typedef float (*StB)[BK * LDB];
    StA As = reinterpret_cast<StA>(smem);
    StB Bs = reinterpret_cast<StB>(reinterpret_cast<float*>(smem) + 2 * BK * LDA);
    if constexpr (!SEG) tile_mainloop<!TA, TB, TILES>(A, lda, B, ldb, m0, n0, kbeg, K, M, N, vecA, vecB, As, Bs, acc);
    else
      for (int sg = 0; sg < nseg; ++sg)
        tile_mainloop<!TA, TB, TILES>(A + sg * segA, lda, B + sg * segB, ldb, m0, n0, kbeg, K, M, N, vecA, vecB, As, Bs, acc, sg == 0);
  }
#ifdef PK2_GEMM_PROFILE
  const long long gp_t1 = wall_clock64();
#endif
  gemm_store<TILES>(acc, M, N, alpha, beta, C, ldc, bias, m0, n0, wm, wn, atomic, act, gate, ldg);
#ifdef PK2_GEMM_PROFILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef PK2_GEMM_PROFILE_N
#define PK2_GEMM_PROFILE_N 4096
#define PK2_GEMM_PROFILE_K 1024
#define PK2_GEMM_PROFILE_EVERY 97
#endif
  if (tid == 0 && K == PK2_GEMM_PROFILE_K && N == PK2_GEMM_PROFILE_N && (blockIdx.x + blockIdx.y * gridDim.x) % PK2_GEMM_PROFILE_EVERY == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    printf("gemm tile (%d,%d) of grid (%d,%d) tiles=%d xcc %u: start %lld main loop %lld epilogue %lld (10 ns ticks)\n", blockIdx.x, blockIdx.y,
           gridDim.x, gridDim.y, TILES, xcc & 7u, gp_t0, gp_t1 - gp_t0, wall_clock64() - gp_t1);
  }
#endif
}

// (Round 6, measured and removed: an XCD-aware tile order -- workgroup l of the dispatch order takes tile (l % 8) (G / 8) + l / 8,
// so that each XCD's L2 sees its own row bands of A instead of all of A -- changed nothing on 20480 x 4096 x 1024 in either
// arithmetic (bf16x3 937 / 947 / 959 us against 905 / 966 / 950 in plain order): the memory-side cache absorbs the 8-fold fetch.)
template <bool TA, bool TB, int TILES, bool X3, bool SEG = false>
__global__ void __launch_bounds__(kGemmThreads, X3 ? PK2_GEMMX_WAVES : 3) gemm_f32_kernel(int M, int N, int K, float alpha,
                                                                const float* __restrict__ A, int64_t lda,
                                                                const float* __restrict__ B, int64_t ldb,
                                                                float beta, float* __restrict__ C, int64_t ldc,
                                                                const float* __restrict__ bias, bool vecA,
                                                                bool vecB, GemmBatch bt) {
  int kbeg = 0;
  {
    int z = blockIdx.z;
    if (bt.ksplit > 1) {
      kbeg = (z % bt.ksplit) * bt.klen;
      z /= bt.ksplit;
      K = min(K, kbeg + bt.klen);
    }
    const int i0 = z / bt.n1, i1 = z % bt.n1;
    A += i0 * bt.sA0 + i1 * bt.sA1;
    B += i0 * bt.sB0 + i1 * bt.sB1;
    C += i0 * bt.sC0 + i1 * bt.sC1;
  }
  __shared__ __attribute__((aligned(16))) char smem[gemm_smem_bytes<TA, TB, TILES, X3>()];
  gemm_block<TA, TB, TILES, X3, SEG>(smem, M, N, K, kbeg, alpha, A, lda, B, ldb, beta, C, ldc, bias, vecA, vecB,
                                blockIdx.y * Geo<TILES>::BMN, blockIdx.x * Geo<TILES>::BMN, bt.ksplit > 1, bt.colsum, bt.act, bt.gate,
                                bt.ldg, bt.nseg, bt.segA, bt.segB);
}

// The row bands of a plain 2-D product in ONE launch (round 4): workgroups [0, nbig) take the 128x128 tiles of rows
// [0, m_split), the others the 64x64 tiles of rows [m_split, M).  As two launches the 64x64 tiles started when the last
// 128x128 tile had finished and ended with a round of their own on a quarter of the CUs (2356 x 4096 x 1024: 125 + 45 us
// + two launch / prologue / epilogue costs = 206 us); here they take the third workgroup slot of a CU from the start and the
// slots the first of them free.
template <bool TA, bool TB, bool X3>
__global__ void __launch_bounds__(kGemmThreads, X3 ? PK2_GEMMX_WAVES : 3) gemm_f32_bands_kernel(int M, int N, int K, float alpha,
                                                                         const float* __restrict__ A, int64_t lda,
                                                                         const float* __restrict__ B, int64_t ldb,
                                                                         float beta, float* __restrict__ C, int64_t ldc,
                                                                         const float* __restrict__ bias, bool vecA,
                                                                         bool vecB, int nbig, int big_cols, int m_split,
                                                                         int small_cols) {
  const int b = blockIdx.x;
  __shared__ __attribute__((aligned(16))) char smem[std::max(gemm_smem_bytes<TA, TB, 2, X3>(), gemm_smem_bytes<TA, TB, 1, X3>())];
  if (b < nbig) {
    gemm_block<TA, TB, 2, X3>(smem, M, N, K, 0, alpha, A, lda, B, ldb, beta, C, ldc, bias, vecA, vecB, (b / big_cols) * 128,
                              (b % big_cols) * 128, false);
  } else {
    const int s = b - nbig;
    gemm_block<TA, TB, 1, X3>(smem, M, N, K, 0, alpha, A, lda, B, ldb, beta, C, ldc, bias, vecA, vecB, m_split + (s / small_cols) * 64,
                              (s % small_cols) * 64, false);
  }
}

// C = beta * C + bias ahead of a split-K product (blockIdx.z = batch entry).
__global__ void __launch_bounds__(256) gemm_prescale_kernel(int M, int N, float beta, float* __restrict__ C,
                                                            int64_t ldc, const float* __restrict__ bias, GemmBatch bt) {
  C += (blockIdx.z / bt.n1) * bt.sC0 + (blockIdx.z % bt.n1) * bt.sC1;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float bv = bias ? bias[n] : 0.f;
  for (int m = blockIdx.y; m < M; m += gridDim.y) {
    float* o = C + (int64_t)m * ldc + n;
    *o = beta == 0.f ? bv : beta * (*o) + bv;
  }
}

// out[n] (+)= sum_m A[m][n].  Rows are split over blockIdx.y; each workgroup reduces its slab of 64 columns
// (4 row partitions of 64 threads, LDS combine) and adds one value per column with a global float atomic.
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ A, int64_t lda, int M, int N,
                                                     int rows_per_block, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float acc = 0.f;
  if (n < N)
    for (int m = m0 + part; m < m1; m += 4) acc += A[(int64_t)m * lda + n];
  red[part][threadIdx.x & 63] = acc;
  __syncthreads();
  if (part == 0 && n < N)
    atomicAdd(out + n, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

__global__ void scale_vec_kernel(float* v, int n, float beta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = beta == 0.f ? 0.f : beta * v[i];
}

}  // namespace pk2

using namespace pk2;

// PK2_GEMM_ARITH = f32 (v_mfma_f32_32x32x2_f32, an f32 fmaf chain) | bf16x3 (three-way bf16 split, six products per k-step
// on v_mfma_f32_32x32x16_bf16, f32 accumulation: gemm_bf16x3.h).  Read once per process; pk2_gemm_set_arith overrides.
#ifndef PK2_GEMM_ARITH_DEFAULT
#define PK2_GEMM_ARITH_DEFAULT 1
#endif
extern "C" int pk2_colsum_f32(const float* A, int64_t lda, int32_t M, int32_t N, float beta, float* out, void* stream_);
static int g_gemm_arith = -1;
static int gemm_arith() {
  if (g_gemm_arith < 0) {
    const char* e = getenv("PK2_GEMM_ARITH");
    g_gemm_arith = e ? (strcmp(e, "f32") == 0 ? 0 : 1) : PK2_GEMM_ARITH_DEFAULT;
  }
  return g_gemm_arith;
}
extern "C" int pk2_gemm_set_arith(int32_t arith) {
  PK2_REQUIRE(arith == 0 || arith == 1, "gemm_set_arith: 0 = f32, 1 = bf16x3");
  g_gemm_arith = arith;
  return PK2_OK;
}
extern "C" int pk2_gemm_get_arith(void) { return gemm_arith(); }

#ifndef PK2_GEMM_MIN_KSLICE
#define PK2_GEMM_MIN_KSLICE 256   // (measured on the layer-0 weight gradient 4096 x 80 x 2276: 101 -> 64 us)
#endif
static int gemm_launch(int transa, int transb, int M, int N, int K, float alpha, const float* A, int64_t lda,
                       const float* B, int64_t ldb, float beta, float* C, int64_t ldc, const float* bias,
                       int n0, const GemmBatch& bt_in, bool aligned_strides, hipStream_t stream) {
  GemmBatch bt = bt_in;
  const bool x3 = gemm_arith() == 1;
  const bool vecA = aligned_strides && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0;
  const bool vecB = aligned_strides && (reinterpret_cast<uintptr_t>(B) & 15) == 0 && (ldb & 3) == 0;
  const int64_t big_tiles = (int64_t)((N + 127) / 128) * ((M + 127) / 128) * n0 * bt.n1;
  const char* force = getenv("PK2_GEMM_TILES");
  // (A stream-K hybrid -- whole tiles in multiples of the CU count, the k-slabs of the rest dealt evenly to further
  // workgroups that add their pieces with atomics -- was built and measured in round 2: once the 64x64-tile kernel kept six
  // slabs in flight behind LDS-only barriers, the row bands / split-K below were faster on every shape of the models, e.g.
  // 2356 x 4096 x 1024: 194 us against 244; removed.  Round 4, after the edge tiles had stopped hiding everything else
  // (gemm_tile.h): a deterministic stream-K -- whole tiles in multiples of the CU count, the k-slabs of the remaining tiles
  // dealt evenly, partial tiles through a write-through workspace, the last contributor of a tile adds the parts in part
  // order -- was parity-green and slower again wherever it applied: 2356 x 512 x 512 25.6 -> 31.1 us, 2356 x 2048 x 512
  // 62.7 -> 74.8, 2356 x 4096 x 1024 203 -> 218 (a second pipeline prologue per workgroup and the partial-tile traffic cost
  // more than the idle tail they remove); not kept.)
  int tiles = force ? atoi(force) : ((big_tiles >= 512 && K >= 256) ? 2 : 1);      // (shallow products: 64x64 tiles, 27 vs 33 us at K = 80)
  // Deep-K products with few output tiles (the weight gradients: K = frames): 128x128 tiles over K slices.
  bt.ksplit = 1;
  bt.klen = K;
  const char* fsplit = getenv("PK2_GEMM_SPLITK");
  // (above one 128x128 tile per CU a plain grid of them gains nothing from slices -- but below two per CU the plain choice
  // is 64x64 tiles at half the intensity: 5768 x 1024 x 20480, the CE output layer's weight gradient, ran at 81 TFLOP/s, 92
  // over three slices, ~125 over eleven; weight-gradient form only -- 2276 x 2560 x 2048 (A W^T) lost 7 % -- PK2_GEMM_SPLIT_MID=0 restores it)
  static const bool split_mid = [] { const char* e = getenv("PK2_GEMM_SPLIT_MID"); return !(e && atoi(e) == 0); }();
  // (how deep K has to be: 2048, but 1536 for a product of at most 16 tiles -- a 512 x 512 weight gradient over K = 1800
  // frames ran on 64 CUs for 43 us, 30 in slices; the TransformerAM has 29 of them per step whenever the minibatch has fewer
  // than 2048 frames.  Measured against it: 48 and 64 tiles at K = 1800 lose 8-10 us in slices, 16 tiles at K = 1000 lose 6)
  const int deep_k = big_tiles <= 16 ? 1536 : 2048;
  // Round 6: the forward form  x W^T  (!transa && transb: every nn.Linear / Conv1d / input projection of a forward pass) is
  // never sliced: slices add with float atomics, i.e. in arrival order, and a forward pass that is reproducible only up to
  // summation order flips ReLU masks of activations within 2e-7 of zero -- one flipped element of a 96 x 2048 mask moved a
  // weight-gradient tensor by 0.2-2 % of its maximum and failed the TransformerAM parity test in 15 % of its runs (round 5
  // ran the tests with PK2_GEMM_SPLITK=1 instead).  Cost at the bench size (FFN down-projection 2300 x 512 x 2048): 57 -> 65 us
  // on 64x64 tiles, 0.1 ms of a 19 ms TransformerAM step; weight gradients and the dX form keep their slices (their
  // summation-order noise of 1e-7 cannot flip anything).  PK2_GEMM_SPLIT_FWD=1 restores the round-5 behaviour.
  static const bool split_fwd = [] { const char* e = getenv("PK2_GEMM_SPLIT_FWD"); return e && atoi(e) == 1; }();
  const bool forward_form = !transa && transb && !split_fwd;
  if (!forward_form && bt.act == 0 && bt.nseg == 1 && (big_tiles <= 256 || (split_mid && transa && !transb && big_tiles < 512)) && K >= deep_k && (!fsplit || atoi(fsplit) != 1)) {
    int ks = (int)std::min<int64_t>((768 + big_tiles - 1) / big_tiles, K / PK2_GEMM_MIN_KSLICE);
    if (big_tiles > 256) {
      // between one and two tiles per CU: the slice count that fills whole rounds of the 512 workgroup slots best (368 tiles:
      // 3 slices leave the last round 28 % idle; measured 2598 us with 3, 2246 with 4, 1998 with 8, 1902 with 16 slices)
      double best = -1.0;
      for (int c = 2; c <= std::min(16, K / 1024); ++c) {
        const int64_t n = big_tiles * c;
        const double eff = (double)n / (double)((n + 511) / 512 * 512);
        if (eff >= best) { best = eff; ks = c; }
      }
    }
    if (fsplit && atoi(fsplit) > 1) ks = atoi(fsplit);
    if (ks > 1 && (int64_t)n0 * bt.n1 * ks <= 65535) {
      tiles = 2;
      bt.klen = ((K + ks - 1) / ks + BK - 1) / BK * BK;
      bt.ksplit = (K + bt.klen - 1) / bt.klen;
      if (beta != 1.f || bias)
        hipLaunchKernelGGL(gemm_prescale_kernel, dim3((N + 255) / 256, std::min(M, 256), n0 * bt.n1), dim3(256), 0,
                           stream, M, N, beta, C, ldc, bias, bt);
    }
  }
  // One launch of `t`-tiles (1 = 64x64, 2 = 128x128) over rows [m_off, m_off + m_rows) of C.
  auto launch = [&](int t, int m_off, int m_rows) {
    const int edge = 64 * t;
    const float* Ap = transa ? A + m_off : A + (int64_t)m_off * lda;
    float* Cp = C + (int64_t)m_off * ldc;
    dim3 grid((N + edge - 1) / edge, (m_rows + edge - 1) / edge, n0 * bt.n1 * bt.ksplit), block(kGemmThreads);
#define PK2_GEMM_T(TA, TB, T, X)                                                                                  \
  hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, T, X>), grid, block, 0, stream, m_rows, N, K, alpha, Ap, lda, B, ldb, \
                     beta, Cp, ldc, bias, vecA, vecB, bt)
#define PK2_GEMM_SEG_T(TB, T, X)                                                                                  \
  hipLaunchKernelGGL((gemm_f32_kernel<false, TB, T, X, true>), grid, block, 0, stream, m_rows, N, K, alpha, Ap, lda, B, ldb, \
                     beta, Cp, ldc, bias, vecA, vecB, bt)
    if (bt.nseg > 1) {       // (A row-major only: pk2_gemm_f32_seg)
      if (transb) { if (t == 2) { if (x3) PK2_GEMM_SEG_T(true, 2, true); else PK2_GEMM_SEG_T(true, 2, false); }
                    else        { if (x3) PK2_GEMM_SEG_T(true, 1, true); else PK2_GEMM_SEG_T(true, 1, false); } }
      else        { if (t == 2) { if (x3) PK2_GEMM_SEG_T(false, 2, true); else PK2_GEMM_SEG_T(false, 2, false); }
                    else        { if (x3) PK2_GEMM_SEG_T(false, 1, true); else PK2_GEMM_SEG_T(false, 1, false); } }
      return;
    }
#define PK2_GEMM(TA, TB)                                                                                          \
  do {                                                                                                            \
    if (t == 2) { if (x3) PK2_GEMM_T(TA, TB, 2, true); else PK2_GEMM_T(TA, TB, 2, false); }                       \
    else        { if (x3) PK2_GEMM_T(TA, TB, 1, true); else PK2_GEMM_T(TA, TB, 1, false); }                       \
  } while (0)
    if (!transa && !transb) PK2_GEMM(false, false);
    else if (!transa && transb) PK2_GEMM(false, true);
    else if (transa && !transb) PK2_GEMM(true, false);
    else PK2_GEMM(true, true);
#undef PK2_GEMM
#undef PK2_GEMM_T
#undef PK2_GEMM_SEG_T
  };
  // Tile quantisation: 128x128 tiles are dealt to 256 CUs, so e.g. the 18 x 32 = 576 tiles of the BLSTM input projection
  // (M = 2276 rows) take three tile-times on some CUs for 2.25 tile-times of work.  A plain 2-D product is therefore cut
  // along M: as many 128-row bands as fill the chip a whole number of times get 128x128 tiles, the remaining rows get
  // 64x64 tiles in a second launch (here: 16 bands = 512 tiles = 2 per CU, then 228 rows x 64 columns-tiles = 256 small
  // tiles = 1 per CU).  Estimated cost of a small tile = 0.3 of a big one (a quarter of the flops at lower intensity).
  int cus = 256;
  const char* band_env = getenv("PK2_GEMM_BANDS");
  const bool no_band = band_env && atoi(band_env) == 0;
  // (between one and two big tiles per CU -- 64x64 tiles throughout -- a band of one big tile per CU plus small ones was
  // slower: 6048 x 1024 x 2356: 351 against 334 us, tools/dbg/gemm_modes.py)
  if (tiles == 2 && !no_band && bt.ksplit == 1 && n0 * bt.n1 == 1 && !force && !bt.colsum && bt.act == 0 && bt.nseg == 1 && (transa ? (lda & 3) == 0 : true)) {
    const int Cn = (N + 127) / 128, R = (M + 127) / 128, Cs = (N + 63) / 64;
    auto rounds = [&](int64_t n) { return (double)((n + cus - 1) / cus); };
    int best_r = R; double best = rounds((int64_t)R * Cn);
    for (int r = 0; r < R; ++r) {
      const int rest = M - r * 128;
      const double c = rounds((int64_t)r * Cn) + 0.3 * rounds((int64_t)((rest + 63) / 64) * Cs);
      if (c < best - 1e-9) { best = c; best_r = r; }
    }
    if (best_r < R) {
      static const bool one_launch = [] { const char* e = getenv("PK2_GEMM_BANDS_ONE"); return !(e && atoi(e) == 0); }();
      if (best_r > 0 && one_launch) {
        const int m_split = best_r * 128, small_rows = (M - m_split + 63) / 64;
        const int nbig = best_r * Cn;
        dim3 grid(nbig + small_rows * Cs), block(kGemmThreads);
#define PK2_GEMM_BANDS_X(TA, TB, X)                                                                                        \
  hipLaunchKernelGGL((gemm_f32_bands_kernel<TA, TB, X>), grid, block, 0, stream, M, N, K, alpha, A, lda, B, ldb, beta, C,   \
                     ldc, bias, vecA, vecB, nbig, Cn, m_split, Cs)
#define PK2_GEMM_BANDS(TA, TB) do { if (x3) PK2_GEMM_BANDS_X(TA, TB, true); else PK2_GEMM_BANDS_X(TA, TB, false); } while (0)
        if (!transa && !transb) PK2_GEMM_BANDS(false, false);
        else if (!transa && transb) PK2_GEMM_BANDS(false, true);
        else if (transa && !transb) PK2_GEMM_BANDS(true, false);
        else PK2_GEMM_BANDS(true, true);
#undef PK2_GEMM_BANDS
#undef PK2_GEMM_BANDS_X
        PK2_LAUNCH_CHECK();
        return PK2_OK;
      }
      if (best_r > 0) launch(2, 0, best_r * 128);
      launch(1, best_r * 128, M - best_r * 128);
      PK2_LAUNCH_CHECK();
      return PK2_OK;
    }
  }
  launch(tiles, 0, M);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

extern "C" int pk2_gemm_f32(int32_t transa, int32_t transb, int32_t M, int32_t N, int32_t K, float alpha,
                            const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                            int64_t ldc, const float* bias, void* stream_) {
  PK2_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "gemm_f32: bad args");
  GemmBatch bt{1, 0, 0, 0, 0, 0, 0, 1, 0};
  return gemm_launch(transa, transb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, 1, bt, true,
                     static_cast<hipStream_t>(stream_));
}

// C = act(alpha op(A) op(B) + beta C + bias): the product with the elementwise pass behind it in its epilogue.  act 1: ReLU (the
// FFN's first Linear); act 2: v kept where gate[m][n] > 0, else 0 -- the ReLU's backward mask read from the forward
// activation (gate may be C's own previous content only with beta = 0: an element is read before it is written by the same
// lane).  Such a product is never cut into atomically added K slices (the activation needs the whole sum).
extern "C" int pk2_gemm_f32_act(int32_t transa, int32_t transb, int32_t M, int32_t N, int32_t K, float alpha,
                                const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                                int64_t ldc, const float* bias, int32_t act, const float* gate, int64_t ldg, void* stream_) {
  PK2_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "gemm_f32_act: bad args");
  PK2_REQUIRE(act == 0 || act == 1 || (act == 2 && gate && ldg >= N), "gemm_f32_act: act is 0, 1 (ReLU) or 2 (gate, ldg >= N)");
  GemmBatch bt{1, 0, 0, 0, 0, 0, 0, 1, 0};
  bt.act = act; bt.gate = gate; bt.ldg = ldg;
  return gemm_launch(transa, transb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, 1, bt, true,
                     static_cast<hipStream_t>(stream_));
}

// C = act(alpha sum_{j < nseg} op(A + j segA) op(B + j segB) + beta C + bias): several products into ONE set of accumulators, one
// launch, no atomics (the sum's order is fixed: bit-reproducible).  TransformerAM's Conv1d(k = 3, pad = 1) over time
// (reference models/transformer.py:88-92) is three taps = three products over row-shifted views of a zero-padded activation
// buffer: segA = +- B*C floats walks the taps' row shifts, segB = C*C the taps' weight slices; K = the depth of ONE segment.
// segA / segB may be negative.  Never cut into K slices.
extern "C" int pk2_gemm_f32_seg(int32_t transa, int32_t transb, int32_t M, int32_t N, int32_t K, int32_t nseg, float alpha,
                                const float* A, int64_t lda, int64_t segA, const float* B, int64_t ldb, int64_t segB, float beta,
                                float* C, int64_t ldc, const float* bias, int32_t act, const float* gate, int64_t ldg,
                                void* stream_) {
  PK2_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && nseg >= 1 && nseg <= 16, "gemm_f32_seg: bad args");
  PK2_REQUIRE(!transa, "gemm_f32_seg: A is row-major [M, K] (transa = 0)");
  PK2_REQUIRE(act == 0 || act == 1 || (act == 2 && gate && ldg >= N), "gemm_f32_seg: act is 0, 1 (ReLU) or 2 (gate, ldg >= N)");
  GemmBatch bt{1, 0, 0, 0, 0, 0, 0, 1, 0};
  bt.act = act; bt.gate = gate; bt.ldg = ldg;
  bt.nseg = nseg; bt.segA = segA; bt.segB = segB;
  // (the vector loads of the fast path need every segment's base 16-byte aligned)
  const bool aligned = nseg == 1 || ((segA & 3) == 0 && (segB & 3) == 0);
  return gemm_launch(transa, transb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, 1, bt, aligned,
                     static_cast<hipStream_t>(stream_));
}

// Weight gradient + bias gradient in one launch: C = alpha A^T B + beta C as pk2_gemm_f32(transa = 1, transb = 0), and
// colsum[m] += sum_k A[k][m] (A is stored [K, M]: the dY of a Linear layer, whose column sums are the bias gradient).  In
// the bf16x3 arithmetic the sums ride in the product's operand loader (float atomics, like the product's K slices); in the
// f32 arithmetic the colsum kernel runs behind the product.
extern "C" int pk2_gemm_f32_tn_colsum(int32_t M, int32_t N, int32_t K, float alpha, const float* A, int64_t lda, const float* B,
                                      int64_t ldb, float beta, float* C, int64_t ldc, float* colsum, void* stream_) {
  PK2_REQUIRE(A && B && C && colsum && M > 0 && N > 0 && K > 0, "gemm_f32_tn_colsum: bad args");
  GemmBatch bt{1, 0, 0, 0, 0, 0, 0, 1, 0};
  static const bool fuse_env = [] { const char* e = getenv("PK2_GEMM_FUSE_COLSUM"); return !(e && atoi(e) == 0); }();
  const bool fuse = fuse_env && gemm_arith() == 1;
  if (fuse) bt.colsum = colsum;
  int rc = gemm_launch(1, 0, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, nullptr, 1, bt, true, static_cast<hipStream_t>(stream_));
  if (rc || fuse) return rc;
  return pk2_colsum_f32(A, lda, K, M, 1.0f, colsum, stream_);
}

extern "C" int pk2_gemm_f32_batched(int32_t transa, int32_t transb, int32_t M, int32_t N, int32_t K, float alpha,
                                    const float* A, int64_t lda, int64_t strideA0, int64_t strideA1,
                                    const float* B, int64_t ldb, int64_t strideB0, int64_t strideB1, float beta,
                                    float* C, int64_t ldc, int64_t strideC0, int64_t strideC1, int32_t n0,
                                    int32_t n1, void* stream_) {
  PK2_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && n0 > 0 && n1 > 0 && (int64_t)n0 * n1 <= 65535,
              "gemm_f32_batched: bad args");
  GemmBatch bt{n1, strideA0, strideA1, strideB0, strideB1, strideC0, strideC1, 1, 0};
  const bool aligned = ((strideA0 | strideA1 | strideB0 | strideB1) & 3) == 0;
  return gemm_launch(transa, transb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, nullptr, n0, bt, aligned,
                     static_cast<hipStream_t>(stream_));
}

extern "C" int pk2_colsum_f32(const float* A, int64_t lda, int32_t M, int32_t N, float beta, float* out,
                              void* stream_) {
  PK2_REQUIRE(A && out && M > 0 && N > 0, "colsum_f32: bad args");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (beta != 1.f) hipLaunchKernelGGL(scale_vec_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, out, N, beta);
  const int col_blocks = (N + 63) / 64;
  const int splits = std::max(1, std::min((M + 63) / 64, 1024 / col_blocks));
  const int rows_per_block = (M + splits - 1) / splits;
  hipLaunchKernelGGL(colsum_kernel, dim3(col_blocks, (M + rows_per_block - 1) / rows_per_block), dim3(256), 0, stream,
                     A, lda, M, N, rows_per_block, out);
  PK2_LAUNCH_CHECK();
  return PK2_OK;
}

// Persistent LSTM recurrence for LARGE batches (B >= 32, H = 512: the CE configuration, 256 chunks x 80 frames) -- gfx950.
//
// The launch-per-step kernels of lstm.hip (lstm_fwd_step_big / lstm_bwd_dh_big + lstm_bwd_pointwise_big) read their 128 KB
// slice of W_hh from memory again in every launch and pay three dependent launches per time step: 17.9 + 17.0 + 5.5 us per
// step against 6.8 us of f32 MFMA work.  Here a whole layer is ONE launch per direction of the data flow:
//  * a TEAM of 32 workgroups (the CUs of one XCD; teams form by arrival order, XCC_ID) owns a task = (direction, tile of
//    64 batch rows): 256 x 80 x 2 directions are exactly 8 tasks for the 8 XCDs, other shapes take turns from a queue;
//  * rank r of a team owns 16 hidden units = 64 gate rows of W_hh and keeps them in LDS (128 KB, laid out so that a lane's
//    B operand of four consecutive 32x32x2 MFMAs is one ds_read_b128) for all time steps;
//  * FORWARD: h_{t-1} of the batch tile (64 x 512) is the A operand; a lane's four k's of four consecutive MFMAs are 16
//    contiguous bytes of a row of y[t-1], loaded straight from the XCD's L2 into registers (eight such loads in flight per
//    lane) -- no staging, no barrier inside the 256-MFMA product; the 64 x 64 gate tile goes through LDS once for the
//    (row, unit) gate math; h is written with agent-scope stores into y[t] itself and announced by one flag word per rank and
//    step (s_waitcnt vmcnt(0) first), which ONE wave per workgroup polls;
//  * BACKWARD: rank r keeps W_hh[:, its 16 units] (all 4H gate rows x 16 columns) in LDS; d h of its units is the product of
//    the batch tile's d gates of the previous step -- d gx[t'] itself, written by the 32 ranks with agent-scope stores and
//    read straight from L2 as the A operand -- with that slice, on 16x16x4 MFMAs; the accumulator leaves d h of (4 rows,
//    one unit) in a lane, where the gate derivatives are computed and stored into d gx[t].  One exchange per step in
//    either direction, through the output tensors themselves.
// Teams / queue / 1 s poll time-out / abort flag as in lstm_persist_seq.hip; a launch that did not finish poisons its output.
// Replaces the same cuDNN recurrence as lstm.hip (reference models/lstm.py:49-58, torch.nn.LSTM).
#include <algorithm>
#include <cstdlib>
#include <map>

#include "gemm_tile.h"
#include "lstm_persist.h"
#include "persist_guard.h"

namespace pk2 {

constexpr int kBgH = 512;
constexpr int kBgR = 32;                    // workgroups of a team = CUs of an XCD; 16 hidden units each
constexpr int kBgTeams = 4;                 // teams per XCD the control block has room for
constexpr int kBgMaxTasks = 64;
constexpr long long kBgSpinTicks = 1000LL * 1000 * 100;     // 1 s of the 100 MHz wall clock
constexpr size_t kBgWFloats = (size_t)64 * kBgH;             // a rank's W_hh slice: 128 KB
constexpr size_t kBgLds = (kBgWFloats + 64 * 65 + 64) * sizeof(float);      // W slice, gate tile (forward), flags

struct BigCtl {
  unsigned arrive[8];
  unsigned next_task;
  unsigned abort;
  unsigned done;
  unsigned pad[5];
  struct Team { unsigned task[kBgMaxTasks + 1]; unsigned bar; unsigned pad[62]; } team[8][kBgTeams];
};

typedef float bg_f32x4 __attribute__((ext_vector_type(4)));

#ifdef PK2_BIG_PROFILE      // phase timers (10 ns wall-clock ticks) of thread 0 of rank 0, summed over the steps of its first task
#define BG_T0() long long bg_last_ = wall_clock64(); long long bg_acc_[6] = {0, 0, 0, 0, 0, 0}
#define BG_T(k) do { const long long n_ = wall_clock64(); bg_acc_[k] += n_ - bg_last_; bg_last_ = n_; } while (0)
#define BG_PRINT(name, steps) do { if (tid == 0 && rank == 0 && iter == 0) printf("%s xcd %d, 10 ns ticks per step over %d steps: %lld | %lld | %lld | %lld | %lld | %lld\n", name, s_i[2], steps, bg_acc_[0] / (steps), bg_acc_[1] / (steps), bg_acc_[2] / (steps), bg_acc_[3] / (steps), bg_acc_[4] / (steps), bg_acc_[5] / (steps)); } while (0)
#else
#define BG_T0() do { } while (0)
#define BG_T(k) do { } while (0)
#define BG_PRINT(name, steps) do { } while (0)
#endif

__device__ __forceinline__ unsigned bg_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0x7;
}
__device__ __forceinline__ unsigned bg_load_u(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bg_store_u(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Data a team hands over inside its XCD goes out with PLAIN stores (round 4): the vector L1 writes through, vmcnt
// acknowledges a store once the XCD's L2 -- the coherence point of the team's 32 CUs -- has it, and the readers' loads
// bypass their L1; an agent-scope (sc1) store is additionally written through the L2 to the fabric on this multi-XCD part
// and its acknowledgement waits for that (tools/ubench/handoff_plain.hip: 0 stale words in 10^11 reads under uneven load;
// lstm_persist_seq.hip: backward step 1.95 -> 1.22 us).  -DPK2_BIG_STOREMODE=1 restores the agent-scope stores.
#ifndef PK2_BIG_STOREMODE
#define PK2_BIG_STOREMODE 0
#endif
__device__ __forceinline__ void bg_store_f(float* p, float v) {
#if PK2_BIG_STOREMODE == 1
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ float bg_sig(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float bg_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }
__device__ __forceinline__ void bg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct BigSpin {
  BigCtl* ctl; long long t0; unsigned n;
  __device__ __forceinline__ explicit BigSpin(BigCtl* c) : ctl(c), t0(0), n(0) {}
  __device__ __forceinline__ bool expired() {
    if ((++n & 255u) != 0u) return false;
    const long long now = wall_clock64();
    if (t0 == 0) t0 = now;
    if (now - t0 > kBgSpinTicks || bg_load_u(&ctl->abort)) {
      bg_store_u(&ctl->abort, 1u);
      return true;
    }
    return false;
  }
};

struct BigRole { int rank; BigCtl::Team* team; };
__device__ __forceinline__ bool bg_register(BigCtl* ctl, int* s_i, BigRole* role) {
  if (threadIdx.x == 0) {
    const unsigned xcd = bg_xcc_id();
    const unsigned slot = __hip_atomic_fetch_add(&ctl->arrive[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_i[0] = (int)(slot % kBgR); s_i[1] = (int)(slot / kBgR); s_i[2] = (int)xcd; s_i[3] = 0;
  }
  __syncthreads();
  if (s_i[1] >= kBgTeams) return false;
  role->rank = s_i[0];
  role->team = &ctl->team[s_i[2]][s_i[1]];
  return true;
}
__device__ __forceinline__ int bg_next_task(BigCtl* ctl, const BigRole& role, int iter, int ntasks, int* s_i) {
  if (threadIdx.x == 0) {
    unsigned k;
    if (role.rank == 0) {
      k = __hip_atomic_fetch_add(&ctl->next_task, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bg_store_u(&role.team->task[iter], k + 1u);
    } else {
      BigSpin spin(ctl);
      unsigned k1;
      while ((k1 = bg_load_u(&role.team->task[iter])) == 0u) {
        if (spin.expired()) { s_i[3] = 1; k1 = 1u << 30; break; }
      }
      k = k1 - 1u;
    }
    s_i[4] = (int)k;
  }
  __syncthreads();
  const int k = s_i[4];
  return (s_i[3] || k >= ntasks) ? -1 : k;
}
// One wave waits until the 32 ranks of the team have announced step `slot`; the others wait at the barrier behind it.
__device__ __forceinline__ bool bg_wait_flags(BigCtl* ctl, const unsigned* flags, int* s_i) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    BigSpin spin(ctl);
    bool ok = true;
    unsigned v = lane < kBgR ? bg_load_u(flags + lane) : 1u;
    while (__ballot(v == 0u) != 0ull) {
      if (spin.expired()) { ok = false; break; }
      if (v == 0u) v = bg_load_u(flags + lane);
    }
    if (!ok && lane == 0) s_i[3] = 1;
  }
  __syncthreads();
  return s_i[3] == 0;
}

// A rank's 64 gate rows of W_hh -> LDS as Wl[k4][n][4]: n = 16 * gate + (unit - u0), the four floats = k 4*k4 .. 4*k4+3.
// (A lane that is column n of a 16x16x4 MFMA and k-phase kq reads, for the chunk of 16 k's c, ONE 16-byte word at
// k4 = 4c + kq and feeds element j of it to the j-th of four MFMAs; the A operand's lane does the same with 16 contiguous
// bytes of h -- the sum over k does not care in which order the MFMAs take the k's.)
__device__ __forceinline__ void bg_load_w(const float* whh_d, int u0, float* Wl) {
  for (int idx = threadIdx.x; idx < 64 * (kBgH / 4); idx += 256) {
    const int n = idx / (kBgH / 4), k4 = idx % (kBgH / 4);
    const int g = n >> 4, u = n & 15;
    const bg_f32x4 v = *reinterpret_cast<const bg_f32x4*>(whh_d + ((size_t)g * kBgH + u0 + u) * kBgH + 4 * k4);
    *reinterpret_cast<bg_f32x4*>(Wl + ((size_t)k4 * 64 + n) * 4) = v;
  }
}

struct BigFwdParams {
  const float* gx;    // [T][B][D*4H]
  const float* whh;   // [D][4H][H]
  const float* bhh;   // [D][4H] or null
  float* y;           // [T][B][D*H]
  float* gates;       // [D][T][B][4H]
  float* cells;       // [D][T][B][H]
  unsigned* flags;    // [ntasks][T][kBgR], zero on entry
  int B, T, D;
};

typedef float bg_f32x4acc __attribute__((ext_vector_type(4)));

// The product of a step is 64 (batch rows) x 64 (16 units x 4 gates) x 512 on 16x16x4 MFMAs: wave w owns the 16-row strip
// 16w .. 16w+15 of the batch tile and all four column tiles -- tile g = gate g of the 16 units -- so every byte of h is read
// ONCE per workgroup (2x2 waves of 32x32 tiles read each half of the batch tile twice: 256 KB per step out of an L2 stream
// that gives a CU 28-34 KB/us, 9 of the step's 12 us).  The tiles are computed TRANSPOSED (the W word is the MFMA's A
// operand, the h word its B operand): the accumulators then leave the four gates of (4 consecutive units, one batch row) in
// one lane -- the gate math runs in those registers, no transposition through LDS, and gx / gates / cells / h move as
// 16-byte words (10 memory instructions per lane and step instead of 40).
__device__ __forceinline__ void bg_store16_agent(float* p, bg_f32x4 v) {
#if PK2_BIG_STOREMODE == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
#endif
}

__global__ void __launch_bounds__(256) lstm_fwd_big_persist(BigFwdParams p, BigCtl* ctl) {
  constexpr int H = kBgH;
  extern __shared__ __attribute__((aligned(16))) float bg_smem[];
  float* Wl = bg_smem;                                   // [H/4][64][4]
  int* s_i = reinterpret_cast<int*>(bg_smem + kBgWFloats + 64 * 65);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  BigRole role;
  if (!bg_register(ctl, s_i, &role)) return;
  const int rank = role.rank, B = p.B, T = p.T, D = p.D;
  const int u0 = 16 * rank;
  const int mtiles = (B + 63) / 64;
  const size_t yrow = (size_t)D * H;
  const int kq = lane >> 4;                              // k-phase of the operand words
  const int uu = u0 + 4 * kq;                            // the lane's results: units uu .. uu+3 of batch row m0 + 16w + (lane & 15)
  for (int iter = 0; iter <= kBgMaxTasks; ++iter) {
    const int task = bg_next_task(ctl, role, iter, D * mtiles, s_i);
    if (task < 0) return;
    const int d = task % D, m0 = (task / D) * 64;
    __syncthreads();                                     // (everybody has left the previous task's LDS)
    bg_load_w(p.whh + (size_t)d * 4 * H * H, u0, Wl);
    bg_f32x4 bias[4], cprev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g)
      bias[g] = p.bhh ? *reinterpret_cast<const bg_f32x4*>(p.bhh + (size_t)d * 4 * H + (size_t)g * H + uu) : bg_f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned* flags = p.flags + (size_t)task * T * kBgR;
    const int b = m0 + 16 * w + (lane & 15);
    const bool live = b < B;
    const int arow = min(b, B - 1);                      // (rows past the batch: a valid row, results never stored)
    const float* wbase = Wl + ((size_t)kq * 64 + (lane & 15)) * 4;
    __syncthreads();
    BG_T0();
    for (int s = 0; s < T; ++s) {
      const int t = d == 0 ? s : T - 1 - s;
      const int tp = d == 0 ? t - 1 : t + 1;
      // gate pre-activations of this step's items: issued now, used behind the matrix product
      bg_f32x4 pre[4];
      {
        const float* gxr = p.gx + ((size_t)t * B + arow) * ((size_t)D * 4 * H) + (size_t)d * 4 * H + uu;
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = *reinterpret_cast<const bg_f32x4*>(gxr + (size_t)g * H);
      }
      BG_T(0);
      bg_f32x4acc acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = bg_f32x4acc{0.f, 0.f, 0.f, 0.f};
      if (s > 0) {
        if (!bg_wait_flags(ctl, flags + (size_t)(s - 1) * kBgR, s_i)) return;
        BG_T(1);
        // W_slice x h_{prev}[strip]^T: 32 chunks of 16 k's, sixteen 16x16x4 MFMAs each; the h words come straight from L2,
        // kDepth chunks ahead; the W words of the next chunk (LDS) are read under this chunk's MFMAs.
        // These are PLAIN loads (a hit in this CU's vector L1 would be stale: other CUs wrote the rows): correct because a
        // line of y[tp] is touched by this CU for the first time only after bg_wait_flags has seen the flags of ALL 32
        // ranks of the step -- every writer of every line of the tile's rows has stored and waited for its stores by
        // then -- and is never written again within the launch, and the L1 is invalidated at the kernel boundary.  A task's
        // rows (direction, 64-row batch tile) are disjoint from every other task's, so a team that takes several tasks
        // never revisits a line either (test_large_batch_lstm_matches_torch_cpu runs B = 600: 20 tasks on 8 teams).  The
        // same holds for the backward kernels' d gates operands.  (agent-scope loads here cost the broadcast
        // out of L2 its L1 hits: the wave pairs of a strip read every word twice.)
        const float* hbase = p.y + ((size_t)tp * B + arow) * yrow + (size_t)d * H + 4 * kq;
        constexpr int kDepth = 8, kChunks = H / 16;
        bg_f32x4 hv[kDepth];
#pragma unroll
        for (int c = 0; c < kDepth; ++c) hv[c] = *reinterpret_cast<const bg_f32x4*>(hbase + 16 * c);
        // (the scheduling barriers keep every load where it is written, kDepth chunks ahead of its use: left alone the
        // scheduler sinks the loads to just before their MFMAs -- two in flight, a round trip to L2 every other chunk)
        __builtin_amdgcn_sched_barrier(0);
        bg_f32x4 wv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) wv[g] = *reinterpret_cast<const bg_f32x4*>(wbase + 64 * g);
        for (int c0 = 0; c0 < kChunks; c0 += kDepth) {
#pragma unroll
          for (int cc = 0; cc < kDepth; ++cc) {
            const int c = c0 + cc;
            const bg_f32x4 hc = hv[cc];
            bg_f32x4 wc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) wc[g] = wv[g];
            const float* wn = wbase + (size_t)min(c + 1, kChunks - 1) * (4 * 64 * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g) wv[g] = *reinterpret_cast<const bg_f32x4*>(wn + 64 * g);
            if (c + kDepth < kChunks) hv[cc] = *reinterpret_cast<const bg_f32x4*>(hbase + 16 * (c + kDepth));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[g].x, hc.x, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[g].y, hc.y, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[g].z, hc.z, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[g].w, hc.w, acc[g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        BG_T(2);
      }
      // accumulator element j of tile g = gate g of (unit uu + j, batch row b)
      bg_f32x4 gi_, gf_, gg_, go_, hh;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ig = bg_sig((pre[0][j] + bias[0][j]) + acc[0][j]), fg = bg_sig((pre[1][j] + bias[1][j]) + acc[1][j]);
        const float gg = bg_tanh((pre[2][j] + bias[2][j]) + acc[2][j]), og = bg_sig((pre[3][j] + bias[3][j]) + acc[3][j]);
        const float c = fg * cprev[j] + ig * gg;
        hh[j] = og * bg_tanh(c);
        cprev[j] = c;
        gi_[j] = ig; gf_[j] = fg; gg_[j] = gg; go_[j] = og;
      }
      if (live) bg_store16_agent(p.y + ((size_t)t * B + b) * yrow + (size_t)d * H + uu, hh);
      BG_T(3);
      // h of this step is in L2 before the flag says so; what only the backward pass reads -- cells, gates -- goes out
      // behind the flag, off the other ranks' critical path
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) bg_store_u(const_cast<unsigned*>(flags) + (size_t)s * kBgR + rank, 1u);
      BG_T(4);
      if (live) {
        *reinterpret_cast<bg_f32x4*>(p.cells + (((size_t)d * T + t) * B + b) * H + uu) = cprev;
        float* gr = p.gates + (((size_t)d * T + t) * B + b) * 4 * H + uu;
        *reinterpret_cast<bg_f32x4*>(gr) = gi_;
        *reinterpret_cast<bg_f32x4*>(gr + (size_t)H) = gf_;
        *reinterpret_cast<bg_f32x4*>(gr + (size_t)2 * H) = gg_;
        *reinterpret_cast<bg_f32x4*>(gr + (size_t)3 * H) = go_;
      }
      BG_T(5);
    }
    BG_PRINT("lstm_fwd_big_persist (gx loads | flag wait | product | gate math + h stores | ack + barrier + flag | cells, gates)", T);
    if (tid == 0 && rank == 0) __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- backward ---------------------------------------------------------------------------------------------------------
// d h[b][k] = d y[t][b][k] + sum_r d gates[t'][b][r] W_hh[r][k] over all 4H gate rows r of the step processed before.  Rank r
// owns the k's of its 16 hidden units and keeps W_hh[:, those 16 columns] (4H x 16 floats = 128 KB) in LDS as
// Wk[r4][n][4]: the four floats = rows 4*r4 .. 4*r4+3 at column u0 + n.  The product is 64 x 16 x 4H on 16x16x4 MFMAs,
// a 16-row strip per wave; the A operand -- the d gates of the whole batch tile -- is d gx[t'] itself, which the 32 ranks
// wrote with agent-scope stores one step earlier: a lane (row, k-phase kq) of the chunk of 16 gate rows c loads the 16
// contiguous bytes r = 16c + 4kq .. +3 of its row straight from L2 and feeds them to four consecutive MFMAs (the B lane
// reads the matching word of Wk).  The accumulator tile leaves a lane with d h of ONE unit for four batch rows: the gate
// derivatives are computed in those registers, no transposition.  (A first version multiplied the rank's own 64 gate rows
// into partial sums for all 512 units and exchanged those: 128 KB out + 128 KB in per rank and step in 64-byte pieces took
// 4.2 + 3.4 us of a 19.8 us step; the all-gather form moves 512 KB in, but in whole lines and under the MFMAs.)
__device__ __forceinline__ void bg_load_w_cols(const float* whh_d, int u0, float* Wk) {
  for (int idx = threadIdx.x; idx < 4 * kBgH * 4; idx += 256) {        // one 16-byte piece (4 of the 16 columns) of a gate row
    const int r = idx >> 2, n4 = idx & 3;
    const bg_f32x4 v = *reinterpret_cast<const bg_f32x4*>(whh_d + (size_t)r * kBgH + u0 + 4 * n4);
    float* dst = Wk + ((size_t)(r >> 2) * 16 + 4 * n4) * 4 + (r & 3);
    dst[0] = v.x; dst[4] = v.y; dst[8] = v.z; dst[12] = v.w;
  }
}

struct BigBwdParams {
  const float* dy;    // [T][B][D*H]
  const float* whh;   // [D][4H][H]
  const float* gates; // [D][T][B][4H]
  const float* cells; // [D][T][B][H]
  float* dgx;         // [T][B][D*4H]
  unsigned* flags;    // [ntasks][T][kBgR], zero on entry
  int B, T, D;
};

__global__ void __launch_bounds__(256) lstm_bwd_big_persist(BigBwdParams p, BigCtl* ctl) {
  constexpr int H = kBgH;
  extern __shared__ __attribute__((aligned(16))) float bg_smem[];
  float* Wk = bg_smem;                                   // [H (4H / 4)][16][4]
  int* s_i = reinterpret_cast<int*>(bg_smem + kBgWFloats);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  BigRole role;
  if (!bg_register(ctl, s_i, &role)) return;
  const int rank = role.rank, B = p.B, T = p.T, D = p.D;
  const int u0 = 16 * rank;
  const int mtiles = (B + 63) / 64;
  const size_t yrow = (size_t)D * H, G4 = 4 * H;
  const int kq = lane >> 4, n = lane & 15;               // MFMA k-phase / output column (own unit) of this lane
  for (int iter = 0; iter <= kBgMaxTasks; ++iter) {
    const int task = bg_next_task(ctl, role, iter, D * mtiles, s_i);
    if (task < 0) return;
    const int d = task % D, m0 = (task / D) * 64;
    __syncthreads();
    bg_load_w_cols(p.whh + (size_t)d * 4 * H * H, u0, Wk);
    const unsigned* flags = p.flags + (size_t)task * T * kBgR;
    const int arow = min(m0 + 16 * w + n, B - 1);        // A operand: this lane's batch row (clamped: results of rows >= B unused)
    const float* bbase = Wk + ((size_t)kq * 16 + n) * 4;
    float dc[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    BG_T0();
    for (int bs = 0; bs < T; ++bs) {
      const int fstep = T - 1 - bs;
      const int t = d == 0 ? fstep : T - 1 - fstep;
      const int tp = d == 0 ? t - 1 : t + 1;             // the forward pass's previous step (c_{t-1})
      const int tn = d == 0 ? t + 1 : t - 1;             // the step processed just before this one
      const bool first_fwd = fstep == 0;
      // the accumulator's rows of this lane: m0 + 16 w + 4 kq + i; forward-pass values of (row, unit n): issued before the wait
      float dyv[4], gi[4], gf[4], gg[4], go[4], cc[4], cp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int b = m0 + 16 * w + 4 * kq + i;
        dyv[i] = gi[i] = gf[i] = gg[i] = go[i] = cc[i] = cp[i] = 0.f;
        if (b < B) {
          dyv[i] = p.dy[((size_t)t * B + b) * yrow + (size_t)d * H + u0 + n];
          const float* gr = p.gates + (((size_t)d * T + t) * B + b) * G4 + u0 + n;
          gi[i] = gr[0]; gf[i] = gr[(size_t)H]; gg[i] = gr[(size_t)2 * H]; go[i] = gr[(size_t)3 * H];
          cc[i] = p.cells[(((size_t)d * T + t) * B + b) * H + u0 + n];
          if (!first_fwd) cp[i] = p.cells[(((size_t)d * T + tp) * B + b) * H + u0 + n];
        }
      }
      BG_T(0);
      bg_f32x4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      if (bs > 0) {
        if (!bg_wait_flags(ctl, flags + (size_t)(bs - 1) * kBgR, s_i)) return;
        BG_T(1);
        const float* abase = p.dgx + ((size_t)tn * B + arow) * ((size_t)D * G4) + (size_t)d * G4 + 4 * kq;
        // (12 words in flight per lane: 34 KB/us per CU, a product of 14.9 us for the 512 KB of a step; 32 in flight: 15.8 us)
        constexpr int kDepth = 12, kChunks = 4 * H / 16;
        bg_f32x4 a[kDepth];
#pragma unroll
        for (int c = 0; c < kDepth; ++c) a[c] = *reinterpret_cast<const bg_f32x4*>(abase + 16 * c);
        bg_f32x4 bv = *reinterpret_cast<const bg_f32x4*>(bbase);
        __builtin_amdgcn_sched_barrier(0);
        for (int c0 = 0; c0 < kChunks; c0 += kDepth) {
#pragma unroll
          for (int cc_ = 0; cc_ < kDepth; ++cc_) {
            const int c = c0 + cc_;
            if (c < kChunks) {
              const bg_f32x4 av = a[cc_], bc = bv;
              bv = *reinterpret_cast<const bg_f32x4*>(bbase + (size_t)min(c + 1, kChunks - 1) * (4 * 16 * 4));
              if (c + kDepth < kChunks) a[cc_] = *reinterpret_cast<const bg_f32x4*>(abase + 16 * (c + kDepth));
              __builtin_amdgcn_sched_barrier(0);
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bc.x, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bc.y, acc1, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bc.z, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bc.w, acc1, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        BG_T(2);
      }
      // gate derivatives of forward step fstep (lstm_bwd_pointwise_big) for (4 rows, unit n), straight from the accumulators
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int b = m0 + 16 * w + 4 * kq + i;
        const float dh = dyv[i] + acc0[i] + acc1[i];
        const float tc = bg_tanh(cc[i]);
        const float dcv = dc[i] + dh * go[i] * (1.f - tc * tc);
        dc[i] = dcv * gf[i];
        if (b < B) {
          float* o = p.dgx + ((size_t)t * B + b) * ((size_t)D * G4) + (size_t)d * G4 + u0 + n;
          bg_store_f(o, dcv * gg[i] * gi[i] * (1.f - gi[i]));
          bg_store_f(o + (size_t)H, dcv * cp[i] * gf[i] * (1.f - gf[i]));
          bg_store_f(o + (size_t)2 * H, dcv * gi[i] * (1.f - gg[i] * gg[i]));
          bg_store_f(o + (size_t)3 * H, dh * tc * go[i] * (1.f - go[i]));
        }
      }
      BG_T(3);
      // this step's d gates are in L2 before the flag says so
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) bg_store_u(const_cast<unsigned*>(flags) + (size_t)bs * kBgR + rank, 1u);
      BG_T(4);
    }
    BG_PRINT("lstm_bwd_big_persist (forward-pass loads | flag wait | product | gate derivatives + d gx stores | ack + barrier + flag | -)", T);
    if (tid == 0 && rank == 0) __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- backward, 4 x 8 decomposition of the team (the default since the end of round 3) ---------------------------------------
// The all-gather form above makes every CU read the batch tile's d gates of ALL 4H gate rows, 512 KB per step, out of an L2
// stream that gives a CU ~34 KB/us: 14.9 of the step's 18.3 us.  Here rank r = (gate g = r / 8, unit group ug = r % 8) keeps
// W_hh[g H .. g H + 511][64 ug .. + 63] (512 x 64 floats, 128 KB) in LDS and multiplies the tile's d gates OF GATE g ONLY
// (64 x 512, 128 KB per step) into a PARTIAL d h for its 64 units -- the same 64 x 64 x 512 product as the forward kernel,
// on the same transposed 16x16x4 tiles.  The four partials of a unit group meet through a 16 KB mailbox per rank in L2
// (agent-scope 16-byte stores, one "partials ready" flag per rank and step); rank (g, ug) then adds them for batch rows
// 16 g .. 16 g + 15, computes the gate derivatives of (16 rows, 64 units) and stores all four gates' d gx with agent-scope
// stores, announced by the step's second flag (which the next step's product waits for on all 32 ranks, as above).
// Two exchanges per step instead of one, a quarter of the operand traffic.
struct BigBwd2Params {
  const float* dy; const float* whh; const float* gates; const float* cells;
  float* dgx;
  float* mail;        // [ntasks][kBgR][64][64]: a rank's partial d h of the step
  unsigned* flags;    // [ntasks][T][kBgR]: d gx of the step stored
  unsigned* pflags;   // [ntasks][T][kBgR]: partials of the step stored
  int B, T, D;
};

__device__ __forceinline__ bg_f32x4 bg_load16_agent(const float* p) {
  bg_f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// W_hh[g H + k][U0 + n], k < 512, n < 64 -> LDS as Wg[k4][n][4]: the four floats = k 4 k4 .. 4 k4 + 3 (the A operand of the
// transposed tiles: a lane that is unit n of a column tile and k-phase kq reads ONE 16-byte word per chunk of 16 k's).
__device__ __forceinline__ void bg_load_w_gate(const float* whh_d, int g, int U0, float* Wg) {
  for (int idx = threadIdx.x; idx < kBgH * 16; idx += 256) {            // one 16-byte piece (4 of the 64 units) of a gate row
    const int k = idx >> 4, n4 = idx & 15;
    const bg_f32x4 v = *reinterpret_cast<const bg_f32x4*>(whh_d + ((size_t)g * kBgH + k) * kBgH + U0 + 4 * n4);
    float* dst = Wg + ((size_t)(k >> 2) * 64 + 4 * n4) * 4 + (k & 3);
    dst[0] = v.x; dst[4] = v.y; dst[8] = v.z; dst[12] = v.w;
  }
}

// One wave waits for the four ranks of its unit group (lanes 0..3 poll rank 8 g' + ug).
__device__ __forceinline__ bool bg_wait_group(BigCtl* ctl, const unsigned* pflags, int ug, int* s_i) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    BigSpin spin(ctl);
    bool ok = true;
    unsigned v = lane < 4 ? bg_load_u(pflags + 8 * lane + ug) : 1u;
    while (__ballot(v == 0u) != 0ull) {
      if (spin.expired()) { ok = false; break; }
      if (v == 0u) v = bg_load_u(pflags + 8 * lane + ug);
    }
    if (!ok && lane == 0) s_i[3] = 1;
  }
  __syncthreads();
  return s_i[3] == 0;
}

__global__ void __launch_bounds__(256) lstm_bwd_big_persist2(BigBwd2Params p, BigCtl* ctl) {
  constexpr int H = kBgH;
  extern __shared__ __attribute__((aligned(16))) float bg_smem[];
  float* Wg = bg_smem;                                   // [H/4][64][4]
  int* s_i = reinterpret_cast<int*>(bg_smem + kBgWFloats + 64 * 65);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  BigRole role;
  if (!bg_register(ctl, s_i, &role)) return;
  const int rank = role.rank, B = p.B, T = p.T, D = p.D;
  const int g = rank >> 3, ug = rank & 7, U0 = 64 * ug;
  const int mtiles = (B + 63) / 64;
  const size_t yrow = (size_t)D * H, G4 = 4 * H;
  const int kq = lane >> 4;
  // the product's lane: batch row 16 w + (lane & 15) of the tile, units 16 nt + 4 kq .. + 3 of the group for nt = 0..3
  // the pointwise thread: batch row 16 g + (tid >> 4), units 4 (tid & 15) .. + 3
  const int prow = 16 * g + (tid >> 4), pu = U0 + 4 * (tid & 15);
  for (int iter = 0; iter <= kBgMaxTasks; ++iter) {
    const int task = bg_next_task(ctl, role, iter, D * mtiles, s_i);
    if (task < 0) return;
    const int d = task % D, m0 = (task / D) * 64;
    __syncthreads();
    bg_load_w_gate(p.whh + (size_t)d * 4 * H * H, g, U0, Wg);
    const unsigned* flags = p.flags + (size_t)task * T * kBgR;
    const unsigned* pflags = p.pflags + (size_t)task * T * kBgR;
    float* mail = p.mail + (size_t)task * kBgR * 64 * 64;
    const int arow = min(m0 + 16 * w + (lane & 15), B - 1);   // (rows past the batch: a valid row, results never used)
    const float* wbase = Wg + ((size_t)kq * 64 + (lane & 15)) * 4;
    const int pb = m0 + prow;
    const bool live = pb < B;
    const int pbc = min(pb, B - 1);
    bg_f32x4 dc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    BG_T0();
    for (int bs = 0; bs < T; ++bs) {
      const int fstep = T - 1 - bs;
      const int t = d == 0 ? fstep : T - 1 - fstep;
      const int tp = d == 0 ? t - 1 : t + 1;             // the forward pass's previous step (c_{t-1})
      const int tn = d == 0 ? t + 1 : t - 1;             // the step processed just before this one
      const bool first_fwd = fstep == 0;
      // forward-pass values of the pointwise items: issued before the waits
      bg_f32x4 dyv, gt[4], cc, cp = {0.f, 0.f, 0.f, 0.f};
      {
        dyv = *reinterpret_cast<const bg_f32x4*>(p.dy + ((size_t)t * B + pbc) * yrow + (size_t)d * H + pu);
        const float* gr = p.gates + (((size_t)d * T + t) * B + pbc) * G4 + pu;
#pragma unroll
        for (int q = 0; q < 4; ++q) gt[q] = *reinterpret_cast<const bg_f32x4*>(gr + (size_t)q * H);
        cc = *reinterpret_cast<const bg_f32x4*>(p.cells + (((size_t)d * T + t) * B + pbc) * H + pu);
        if (!first_fwd) cp = *reinterpret_cast<const bg_f32x4*>(p.cells + (((size_t)d * T + tp) * B + pbc) * H + pu);
      }
      BG_T(0);
      bg_f32x4 dh = dyv;
      if (bs > 0) {
        if (!bg_wait_flags(ctl, flags + (size_t)(bs - 1) * kBgR, s_i)) return;
        BG_T(1);
        // W_gate x (d gates of gate g)^T: 32 chunks of 16 k's, sixteen 16x16x4 MFMAs each (as the forward kernel)
        const float* abase = p.dgx + ((size_t)tn * B + arow) * ((size_t)D * G4) + (size_t)d * G4 + (size_t)g * H + 4 * kq;
        constexpr int kDepth = 8, kChunks = H / 16;
        bg_f32x4 av[kDepth];
#pragma unroll
        for (int c = 0; c < kDepth; ++c) av[c] = *reinterpret_cast<const bg_f32x4*>(abase + 16 * c);
        __builtin_amdgcn_sched_barrier(0);
        bg_f32x4 wv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[q] = *reinterpret_cast<const bg_f32x4*>(wbase + 64 * q);
        bg_f32x4acc acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = bg_f32x4acc{0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < kChunks; c0 += kDepth) {
#pragma unroll
          for (int cc_ = 0; cc_ < kDepth; ++cc_) {
            const int c = c0 + cc_;
            const bg_f32x4 ac = av[cc_];
            bg_f32x4 wc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wc[q] = wv[q];
            const float* wn = wbase + (size_t)min(c + 1, kChunks - 1) * (4 * 64 * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = *reinterpret_cast<const bg_f32x4*>(wn + 64 * q);
            if (c + kDepth < kChunks) av[cc_] = *reinterpret_cast<const bg_f32x4*>(abase + 16 * (c + kDepth));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[q].x, ac.x, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[q].y, ac.y, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[q].z, ac.z, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[q].w, ac.w, acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        BG_T(2);
        // the partial of (row 16 w + (lane & 15), units 16 q + 4 kq .. + 3) -> this rank's mailbox
        {
          float* mrow = mail + ((size_t)rank * 64 + 16 * w + (lane & 15)) * 64 + 4 * kq;
#pragma unroll
          for (int q = 0; q < 4; ++q) bg_store16_agent(mrow + 16 * q, bg_f32x4{acc[q][0], acc[q][1], acc[q][2], acc[q][3]});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) bg_store_u(const_cast<unsigned*>(pflags) + (size_t)bs * kBgR + rank, 1u);
        if (!bg_wait_group(ctl, pflags + (size_t)bs * kBgR, ug, s_i)) return;
        BG_T(3);
#pragma unroll
        for (int q = 0; q < 4; ++q) {                      // (always in the order of the gates: the sum is the same in every rank)
          const bg_f32x4 pv = bg_load16_agent(mail + ((size_t)(8 * q + ug) * 64 + prow) * 64 + 4 * (tid & 15));
          dh += pv;
        }
      }
      // gate derivatives of forward step fstep for (row prow, units pu .. pu + 3)
      bg_f32x4 di, df, dg, dO;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float tc = bg_tanh(cc[j]);
        const float dcv = dc[j] + dh[j] * gt[3][j] * (1.f - tc * tc);
        dc[j] = dcv * gt[1][j];
        di[j] = dcv * gt[2][j] * gt[0][j] * (1.f - gt[0][j]);
        df[j] = dcv * cp[j] * gt[1][j] * (1.f - gt[1][j]);
        dg[j] = dcv * gt[0][j] * (1.f - gt[2][j] * gt[2][j]);
        dO[j] = dh[j] * tc * gt[3][j] * (1.f - gt[3][j]);
      }
      if (live) {
        float* o = p.dgx + ((size_t)t * B + pb) * ((size_t)D * G4) + (size_t)d * G4 + pu;
        bg_store16_agent(o, di);
        bg_store16_agent(o + (size_t)H, df);
        bg_store16_agent(o + (size_t)2 * H, dg);
        bg_store16_agent(o + (size_t)3 * H, dO);
      }
      BG_T(4);
      // this step's d gates are in L2 before the flag says so
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) bg_store_u(const_cast<unsigned*>(flags) + (size_t)bs * kBgR + rank, 1u);
      BG_T(5);
    }
    BG_PRINT("lstm_bwd_big_persist2 (forward-pass loads | flag wait | product | partials out + group wait | partials in + gate derivatives + d gx stores | ack + barrier + flag)", T);
    if (tid == 0 && rank == 0) __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// A launch in which a poll timed out (or a task was never finished) must not pass for a result.
__global__ void lstm_big_check(const BigCtl* ctl, unsigned ntasks, float* out, size_t n, unsigned* sticky, unsigned* guard_dev, unsigned* guard_host) {
  if (ctl->abort == 0u && ctl->done == ntasks) return;
  if (threadIdx.x == 0) { *sticky = 1u; persist_guard_raise(guard_dev, guard_host); }
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) out[i] = __uint_as_float(0x7fc00000u);
}

// ---- host -------------------------------------------------------------------------------------------------------------
struct BigScratch { BigCtl* ctl = nullptr; unsigned* flags = nullptr; size_t flag_words = 0; unsigned* sticky = nullptr;
                    float* mail = nullptr; size_t mail_floats = 0; PersistGuard guard; };
static std::map<DevStream, BigScratch> g_big_scratch;
static PerDevice<int> g_big_state_pd(-1);             // -1 untested, 0 unusable, 1 verified on this device

static int big_scratch(hipStream_t stream, size_t flag_words, BigScratch** out) {
  BigScratch& sc = g_big_scratch[dev_stream(stream)];
  if (!sc.ctl) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.ctl), sizeof(BigCtl)));
  if (!sc.sticky) {
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.sticky), sizeof(unsigned)));
    PK2_HIP(hipMemsetAsync(sc.sticky, 0, sizeof(unsigned), stream));
  }
  if (!sc.guard.dev) { int rc = persist_guard(&sc.guard); if (rc) return rc; }
  if (sc.flag_words < flag_words) {
    if (sc.flags) PK2_HIP(hipFree(sc.flags));
    sc.flags = nullptr; sc.flag_words = 0;
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.flags), flag_words * sizeof(unsigned)));
    sc.flag_words = flag_words;
  }
  *out = &sc;
  return PK2_OK;
}

bool lstm_big_wanted(int B, int H, int D) {
  const char* env = getenv("PK2_LSTM_BIG_PERSIST");
  if (env && atoi(env) == 0) return false;
  if (g_big_state_pd.ref() == 0 || H != kBgH || B < 32 || (D != 1 && D != 2) || D * ((B + 63) / 64) > kBgMaxTasks) return false;
  static PerDevice<int> cus_pd(-1); int& cus = cus_pd.ref();
  if (cus < 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus = n;
  }
  return cus == 8 * kBgR;
}

int lstm_fwd_big_launch(const float* gx, const float* whh, const float* bhh, int B, int T, int H, int D, float* y,
                        float* gates, float* cells, hipStream_t stream, bool* ran) {
  *ran = false;
  // (the kernel moves gx / h / cells / gates / the bias as 16-byte words: tensors that are not 16-byte aligned keep the
  // step kernels)
  if (((reinterpret_cast<uintptr_t>(gx) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gates) |
        reinterpret_cast<uintptr_t>(cells) | reinterpret_cast<uintptr_t>(bhh)) & 15) != 0) return PK2_OK;
  const int ntasks = D * ((B + 63) / 64);
  BigScratch* sc = nullptr;
  int rc = big_scratch(stream, (size_t)ntasks * T * kBgR, &sc);
  if (rc) return rc;
  PK2_HIP(hipMemsetAsync(sc->ctl, 0, sizeof(BigCtl), stream));
  PK2_HIP(hipMemsetAsync(sc->flags, 0, (size_t)ntasks * T * kBgR * sizeof(unsigned), stream));
  static PerDevice<bool> attr_pd(false); bool& attr = attr_pd.ref();
  if (!attr) {
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_big_persist), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr = true;
  }
  BigFwdParams p{gx, whh, bhh, y, gates, cells, sc->flags, B, T, D};
  hipLaunchKernelGGL(lstm_fwd_big_persist, dim3(8 * kBgR), dim3(256), kBgLds, stream, p, sc->ctl);
  PK2_LAUNCH_CHECK();
  if (g_big_state_pd.ref() < 0) {                 // first use on this device: every task done, nobody timed out?
    BigCtl* h = new BigCtl;
    hipError_t e = hipMemcpyAsync(h, sc->ctl, sizeof(BigCtl), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    const bool ok = e == hipSuccess && h->abort == 0u && h->done == (unsigned)ntasks;
    delete h;
    if (e != hipSuccess) { set_error("lstm_big: %s", hipGetErrorString(e)); return PK2_ERR_HIP; }
    g_big_state_pd.ref() = ok ? 1 : 0;
    if (!ok) return PK2_OK;              // the caller falls back (and keeps doing so)
  }
  hipLaunchKernelGGL(lstm_big_check, dim3(1), dim3(1024), 0, stream, sc->ctl, (unsigned)ntasks, y, (size_t)T * B * D * H, sc->sticky, sc->guard.dev, sc->guard.host_dev);
  *ran = true;
  return PK2_OK;
}

int lstm_bwd_big_launch(const float* dy, const float* whh, const float* gates, const float* cells, int B, int T, int H,
                        int D, float* dgx, hipStream_t stream, bool* ran) {
  *ran = false;
  if (g_big_state_pd.ref() != 1) return PK2_OK;   // the forward pass verifies the device first
  const int ntasks = D * ((B + 63) / 64);
  BigScratch* sc = nullptr;
  int rc = big_scratch(stream, (size_t)2 * ntasks * T * kBgR, &sc);      // "d gx stored" flags, then "partials stored" flags
  if (rc) return rc;
  PK2_HIP(hipMemsetAsync(sc->ctl, 0, sizeof(BigCtl), stream));
  PK2_HIP(hipMemsetAsync(sc->flags, 0, (size_t)2 * ntasks * T * kBgR * sizeof(unsigned), stream));
  static PerDevice<bool> attr_pd(false); bool& attr = attr_pd.ref();
  if (!attr) {
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_big_persist), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr = true;
  }
  // PK2_LSTM_BIG_BWD=1: the all-gather form (every rank reads all 4H d gates); default: the 4 x 8 decomposition
  static const bool form2 = [] { const char* e = getenv("PK2_LSTM_BIG_BWD"); return !(e && atoi(e) == 1); }();
  const bool aligned = ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(gates) | reinterpret_cast<uintptr_t>(cells) |
                         reinterpret_cast<uintptr_t>(dgx)) & 15) == 0;
  if (form2 && aligned) {
    const size_t mail_floats = (size_t)ntasks * kBgR * 64 * 64;
    if (sc->mail_floats < mail_floats) {
      if (sc->mail) PK2_HIP(hipFree(sc->mail));
      sc->mail = nullptr; sc->mail_floats = 0;
      PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc->mail), mail_floats * sizeof(float)));
      sc->mail_floats = mail_floats;
    }
    static PerDevice<bool> attr2_pd(false); bool& attr2 = attr2_pd.ref();
    if (!attr2) {
      PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_big_persist2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024));
      attr2 = true;
    }
    BigBwd2Params p2{dy, whh, gates, cells, dgx, sc->mail, sc->flags, sc->flags + (size_t)ntasks * T * kBgR, B, T, D};
    hipLaunchKernelGGL(lstm_bwd_big_persist2, dim3(8 * kBgR), dim3(256), kBgLds, stream, p2, sc->ctl);
  } else {
    BigBwdParams p{dy, whh, gates, cells, dgx, sc->flags, B, T, D};
    hipLaunchKernelGGL(lstm_bwd_big_persist, dim3(8 * kBgR), dim3(256), kBgLds, stream, p, sc->ctl);
  }
  hipLaunchKernelGGL(lstm_big_check, dim3(1), dim3(1024), 0, stream, sc->ctl, (unsigned)ntasks, dgx, (size_t)T * B * D * 4 * H, sc->sticky, sc->guard.dev, sc->guard.host_dev);
  PK2_LAUNCH_CHECK();
  *ran = true;
  return PK2_OK;
}

int lstm_big_status(unsigned* abort_flag) {
  unsigned any = 0;
  for (auto& kv : g_big_scratch) {
    unsigned st = 0;
    if (kv.second.sticky && hipMemcpy(&st, kv.second.sticky, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess) any |= st;
  }
  *abort_flag = any;
  return PK2_OK;
}

}  // namespace pk2

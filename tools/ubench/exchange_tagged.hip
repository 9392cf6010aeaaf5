// Micro-benchmark (priced in round 6): the per-frame all-gather of the persistent denominator kernel -- 32
// workgroups of an XCD each rewrite their slice of a state vector (120 KB on the bench graph) and then every one of them needs
// the WHOLE vector in its LDS -- under two protocols:
//   A  (what den_persist2_kernel does)  plain slice stores -> s_waitcnt vmcnt(0) -> agent-scope word per rank -> one wave polls
//      the 32 words -> barrier -> LDS-DMA copy of the vector -> wait -> barrier
//   B  (tagged data)  the values carry their frame's parity in the SIGN bit (an alpha is never negative), three ring slots so
//      that a slot's previous tenant has the other parity; slice stores -> at once the LDS-DMA copy of the vector -> wait ->
//      barrier -> every thread checks the sign of its share of the LDS copy, rows with a stale granule are copied again and
//      checked again (no word, no wait for the own stores' acknowledgement in front of a publication)
// with W microseconds of stand-in work per frame (s_sleep) whose length differs by rank (ranks arrive within ~0.1 us of each
// other in the kernel; here: +-`skew` ns, rotating).  Output: us per frame, retries per frame (B).
//   hipcc --offload-arch=gfx950 -O3 -o exchange_tagged exchange_tagged.hip && ./exchange_tagged
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kT = 512, kW = kT / 64, kR = 32;
constexpr unsigned kEmptyWord = 0xffffffffu;

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0x7;
}
__device__ __forceinline__ void dma_row(const float* g_lane, float* lds_wave_base) {
  const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1" : : "s"(m0v), "v"(g_lane) : "memory");
}
struct Ctl { unsigned reg[9]; unsigned fail; unsigned words[8][3][kR]; unsigned long long retries; };

__device__ __forceinline__ void work(int ns) {       // stand-in for the passes: ~ns nanoseconds
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ns / 10) __builtin_amdgcn_s_sleep(1);
}

__global__ void __launch_bounds__(kT) k(float* ring /* [8][3][floats] */, int floats, int mode, int frames, int work_ns, int skew_ns,
                                        Ctl* c, unsigned long long* ticks, unsigned* bad) {
  extern __shared__ float lds[];
  __shared__ int s_slot, s_members, s_abort;
  __shared__ unsigned s_badrows[4];       // 128 rows
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned xcd = xcc_id();
  if (tid == 0) {
    s_abort = 0;
    s_slot = (int)__hip_atomic_fetch_add(&c->reg[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&c->reg[8], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&c->reg[8], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x)
      if (++spins > 50000000u) { c->fail = 1u; break; }
    s_members = (int)__hip_atomic_load(&c->reg[xcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int rank = s_slot, members = s_members;
  if (members != kR) return;
  const int rows = (floats + 255) / 256;
  const int slice = (floats + kR - 1) / kR, s0 = rank * slice, s1 = min(floats, s0 + slice);
  float* base = ring + (size_t)xcd * 3 * floats;
  unsigned long long acc = 0, retries = 0;
  for (int t = 1; t <= frames; ++t) {
    const long long t_in = wall_clock64();
    work(work_ns + ((rank + t) % 5) * skew_ns / 4);
    float* v = base + (size_t)(t % 3) * floats;
    const float sign = (mode == 1 && (t & 1)) ? -1.f : 1.f;
    for (int i = s0 + tid; i < s1; i += kT) v[i] = sign * (float)(1 + ((i * 7 + t) & 1023));
    if (mode == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(&c->words[xcd][t % 3][rank], (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&c->words[xcd][(t + 1) % 3][rank], kEmptyWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (w == 0 && lane < kR) {
        unsigned spins = 0;
        while (__hip_atomic_load(&c->words[xcd][t % 3][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)t)
          if (++spins > 20000000u) { s_abort = 1; break; }
      }
      __syncthreads();
      if (s_abort) { if (tid == 0) c->fail = 2u; return; }
      for (int r = w; r < rows; r += kW) dma_row(v + (size_t)r * 256 + lane * 4, lds + (size_t)r * 256);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      for (int r = w; r < rows; r += kW) dma_row(v + (size_t)r * 256 + lane * 4, lds + (size_t)r * 256);
      unsigned rounds = 0;
      while (true) {
        if (tid < 4) s_badrows[tid] = 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // a granule of 4 floats per lane and step: stale if any of its signs is the other frame's
        unsigned m[4] = {0u, 0u, 0u, 0u};          // (per thread, then per wave: one LDS atomic per wave and word)
        for (int gidx = tid; gidx * 4 < floats; gidx += kT) {
          const float4 q = *reinterpret_cast<const float4*>(lds + (size_t)gidx * 4);
          const bool neg = (t & 1) != 0;
          const bool ok = neg ? (q.x < 0.f && q.y < 0.f && q.z < 0.f && q.w < 0.f) : (q.x > 0.f && q.y > 0.f && q.z > 0.f && q.w > 0.f);
          const int row = gidx >> 6;
          if (!ok) {
            const unsigned bit = 1u << (row & 31);
            if (row < 32) m[0] |= bit; else if (row < 64) m[1] |= bit; else if (row < 96) m[2] |= bit; else m[3] |= bit;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned v = m[q];
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
          if (lane == 0 && v) atomicOr(&s_badrows[q], v);
        }
        __syncthreads();
        const unsigned b0 = s_badrows[0], b1 = s_badrows[1], b2 = s_badrows[2], b3 = s_badrows[3];
        if ((b0 | b1 | b2 | b3) == 0u) break;
        if (++rounds > 200000u) { if (tid == 0) c->fail = 3u; return; }
        __syncthreads();
        for (int r = w; r < rows; r += kW) {
          const unsigned word = r < 32 ? b0 : r < 64 ? b1 : r < 96 ? b2 : b3;
          if ((word >> (r & 31)) & 1u) dma_row(v + (size_t)r * 256 + lane * 4, lds + (size_t)r * 256);
        }
      }
      retries += rounds;
    }
    // every word must be this frame's
    if ((t & 63) == 0) {
      unsigned wrong = 0;
      for (int i = tid; i < floats; i += kT) wrong += fabsf(lds[i]) != (float)(1 + ((i * 7 + t) & 1023));
      if (wrong) atomicAdd(bad, wrong);
    }
    acc += (unsigned long long)(wall_clock64() - t_in);
  }
  if (tid == 0) { ticks[blockIdx.x] = acc; atomicAdd(&c->retries, retries); }
}

int main() {
  const int frames = 2000, floats = 120 * 256;
  float* ring; Ctl* c; unsigned long long* ticks; unsigned* bad;
  CK(hipMalloc(&ring, (size_t)8 * 3 * floats * 4)); CK(hipMalloc(&c, sizeof(Ctl))); CK(hipMalloc(&ticks, 256 * 8)); CK(hipMalloc(&bad, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 124 * 1024));
  printf("# us per frame (stand-in work included), 32 workgroups per XCD, 8 XCDs, 120 KB vector, %d frames\n", frames);
  printf("%-10s %-10s %12s %12s %10s\n", "work ns", "skew ns", "A: words", "B: tagged", "B retries");
  for (int work_ns : {0, 3000}) for (int skew_ns : {0, 200, 800}) {
    double us[2]; double rt = 0;
    for (int mode = 0; mode < 2; ++mode) {
      CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipMemset(ticks, 0, 256 * 8)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(ring, 0, (size_t)8 * 3 * floats * 4));
      hipLaunchKernelGGL(k, dim3(256), dim3(kT), 124 * 1024, 0, ring, floats, mode, frames, work_ns, skew_ns, c, ticks, bad);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h(256); unsigned hb = 0; Ctl hc;
      CK(hipMemcpy(h.data(), ticks, 256 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost));
      double sum = 0; int n = 0;
      for (auto t : h) if (t) { sum += (double)t; ++n; }
      us[mode] = (hb || hc.fail || !n) ? -(double)(hc.fail ? hc.fail : 9) : sum / n / frames * 0.01;
      if (mode == 1) rt = (double)hc.retries / 256.0 / frames;
    }
    printf("%-10d %-10d %12.3f %12.3f %10.3f\n", work_ns, skew_ns, us[0], us[1], rt);
  }
  return 0;
}

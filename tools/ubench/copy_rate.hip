// Micro-benchmark (VERDICT r5 next #2a): how fast can ONE CU bring a state vector that the 32 workgroups of its XCD have just
// rewritten (each its 1/32 slice, as the persistent denominator kernel's ranks do every frame) from the XCD's L2 into its LDS?
//   mode 0  LDS-DMA            global_load_lds_dwordx4 sc1, 1 KB rows dealt to the 8 waves round robin (what the kernel does)
//   mode 1  register path      per lane G x global_load_dwordx4 sc1 in flight, then G x ds_write_b128
//   mode 2  mixed              waves 0-3 LDS-DMA, waves 4-7 register path, rows split evenly
// x readers per XCD in {1, 8, 32} (the other workgroups of the XCD still rewrite their slices and wait at the barrier)
// x bytes in {60 KB (a table chunk of the bench graph), 120 KB (its whole vector)}.
// Output: KB/us per CU (copy issue -> data in LDS, s_waitcnt vmcnt(0) + barrier), mean over readers and iterations.
//   hipcc --offload-arch=gfx950 -O3 -o copy_rate copy_rate.hip && ./copy_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kT = 512, kW = kT / 64;
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0x7;
}
__device__ __forceinline__ void dma_row(const float* g_lane, float* lds_wave_base) {
  const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1" : : "s"(m0v), "v"(g_lane) : "memory");
}
__device__ __forceinline__ f4 load16_sc1(const float* p) {
  f4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct Ctl { unsigned reg[9]; unsigned bar[8]; unsigned fail; };

// XCD barrier: monotonic counter per XCD, release before arrive, relaxed sc1 poll, acquire after.
__device__ __forceinline__ bool xcd_barrier(Ctl* c, unsigned xcd, unsigned members, unsigned* gen) {
  __syncthreads();
  const unsigned target = members * ++*gen;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&c->bar[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&c->bar[xcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 50000000u) { __hip_atomic_store(&c->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
  return __hip_atomic_load(&c->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
}

template <int G>
__global__ void __launch_bounds__(kT) k(float* vec /* [8][floats] */, int floats, int mode, int readers, int iters, Ctl* c,
                                        unsigned long long* ticks /* [256] */, unsigned* bad) {
  extern __shared__ float lds[];
  __shared__ int s_slot, s_members;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned xcd = xcc_id();
  if (tid == 0) {
    s_slot = (int)__hip_atomic_fetch_add(&c->reg[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&c->reg[8], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&c->reg[8], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x)
      if (++spins > 50000000u) { c->fail = 1u; break; }
    s_members = (int)__hip_atomic_load(&c->reg[xcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int slot = s_slot, members = s_members;
  float* v = vec + (size_t)xcd * floats;
  const int slice = (floats + members - 1) / members, s0 = slot * slice, s1 = min(floats, s0 + slice);
  const int rows = (floats + 255) / 256;            // 1 KB rows
  unsigned gen = 0;
  unsigned long long acc = 0;
  for (int it = 1; it <= iters; ++it) {
    // every member rewrites its slice (plain stores; the barrier's release fence publishes them)
    for (int i = s0 + tid; i < s1; i += kT) v[i] = (float)((i * 7 + it) & 1023);
    if (!xcd_barrier(c, xcd, (unsigned)members, &gen)) return;
    if (slot < readers) {
      const long long t0 = wall_clock64();
      if (mode == 0) {
        for (int r = w; r < rows; r += kW) dma_row(v + (size_t)r * 256 + lane * 4, lds + (size_t)r * 256);
      } else {
        const int w0 = mode == 2 ? 4 : 0, nw = kW - w0;
        const int r_dma = mode == 2 ? rows / 2 : 0;
        if (w < w0) {
          for (int r = w; r < r_dma; r += w0) dma_row(v + (size_t)r * 256 + lane * 4, lds + (size_t)r * 256);
        } else {
          // rows [r_dma, rows): lane i of wave (w - w0) takes granules (r * 64 + lane), r = w - w0, w - w0 + nw, ...; G in flight
          int r = r_dma + (w - w0);
          while (r < rows) {
            f4 x[G]; int rr[G];
#pragma unroll
            for (int g = 0; g < G; ++g) { rr[g] = r + g * nw; if (rr[g] < rows) x[g] = load16_sc1(v + (size_t)rr[g] * 256 + lane * 4); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int g = 0; g < G; ++g) if (rr[g] < rows) *reinterpret_cast<f4*>(lds + (size_t)rr[g] * 256 + lane * 4) = x[g];
            r += G * nw;
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      const long long t1 = wall_clock64();
      if (tid == 0) acc += (unsigned long long)(t1 - t0);
      // every word must be this iteration's
      unsigned wrong = 0;
      for (int i = tid; i < floats; i += kT) wrong += lds[i] != (float)((i * 7 + it) & 1023);
      if (wrong) atomicAdd(bad, wrong);
    }
    if (!xcd_barrier(c, xcd, (unsigned)members, &gen)) return;      // nobody rewrites before every reader is done
  }
  if (tid == 0 && slot < readers) ticks[blockIdx.x] = acc;
}

int main() {
  const int iters = 200;
  float* vec; Ctl* c; unsigned long long* ticks; unsigned* bad;
  const int maxf = 120 * 256;
  CK(hipMalloc(&vec, (size_t)8 * maxf * 4)); CK(hipMalloc(&c, sizeof(Ctl))); CK(hipMalloc(&ticks, 256 * 8)); CK(hipMalloc(&bad, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 124 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 124 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<15>), hipFuncAttributeMaxDynamicSharedMemorySize, 124 * 1024));
  printf("# KB/us per CU, data rewritten by the XCD's 32 workgroups before every copy; 512 threads, 1 workgroup per CU, all 8 XCDs\n");
  printf("%-34s %8s %8s %8s\n", "mode / bytes", "1 reader", "8", "32");
  const char* names[] = {"LDS-DMA", "register path", "mixed 4+4 waves"};
  for (int kb : {60, 120}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int G : {4, 8, 15}) {
        if (mode == 0 && G != 8) continue;
        double out[3];
        int col = 0;
        for (int readers : {1, 8, 32}) {
          CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipMemset(ticks, 0, 256 * 8)); CK(hipMemset(bad, 0, 4));
          const int floats = kb * 256;
          if (G == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(kT), 124 * 1024, 0, vec, floats, mode, readers, iters, c, ticks, bad);
          else if (G == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(kT), 124 * 1024, 0, vec, floats, mode, readers, iters, c, ticks, bad);
          else hipLaunchKernelGGL(k<15>, dim3(256), dim3(kT), 124 * 1024, 0, vec, floats, mode, readers, iters, c, ticks, bad);
          CK(hipDeviceSynchronize());
          std::vector<unsigned long long> h(256); unsigned hb = 0; Ctl hc;
          CK(hipMemcpy(h.data(), ticks, 256 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
          CK(hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost));
          double sum = 0; int n = 0;
          for (auto t : h) if (t) { sum += (double)t; ++n; }
          const double us = n ? sum / n / iters * 0.01 : 0.0;       // wall_clock64: 10 ns ticks
          out[col++] = (hb || hc.fail || !n) ? -1.0 : kb / us;
        }
        printf("%-16s G=%-2d %4d KB        %8.1f %8.1f %8.1f\n", names[mode], G, kb, out[0], out[1], out[2]);
      }
    }
  }
  return 0;
}

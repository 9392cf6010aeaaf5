// Micro-benchmark (development aid): price of one frame of an LDS-resident denominator recursion on MI355X.
// A persistent kernel keeps its slice of the arc list in LDS (8-byte records {state, prob}) for all frames;
// per frame every workgroup gathers 16-byte (forward half) / 32-byte (backward half) per-state vectors from
// the previous frame, reduces them into rows, writes its rows of the next frame and crosses a grid barrier
// (agent-scope release -> counter -> acquire).  Compare with the launch-per-frame kernel (24 us / frame).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kArcsPerHalf = 4096;   // per workgroup and direction
constexpr int kK = 8;
constexpr int kRows = 128;

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* fail) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 4000000u) { atomicAdd(fail, 1u); ok = false; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

// Flag barrier: every workgroup publishes the step in its own slot; 256 threads poll the 256 slots.
template <bool FENCE>
__device__ __forceinline__ bool flag_barrier(unsigned* flags, unsigned step, unsigned* fail) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store(flags + blockIdx.x, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  bool ok = true;
  if (threadIdx.x < gridDim.x) {
    unsigned spins = 0;
    while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < step) {
      if (++spins > 4000000u) { atomicAdd(fail, 1u); ok = false; break; }
    }
  }
  if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  ok = __syncthreads_and(ok);
  return ok;
}

template <int MODE>   // bit 0: gathers/reduce/write; bits 1-2: 0 counter barrier, 1 flag barrier, 2 flag barrier + fences, 3 none
__global__ void __launch_bounds__(1024) frames(const int2* __restrict__ arcs, float* a0, float* a1, float* b0, float* b1,
                                               int S, int steps, unsigned* ctr, unsigned* fail, float* parts, unsigned* flags) {
  extern __shared__ int2 lds_arcs[];                 // [2][kArcsPerHalf]
  __shared__ float acc[2][kRows * 4];
  const int tid = threadIdx.x, wg = blockIdx.x, half = tid >> 9, ht = tid & 511;
  for (int i = tid; i < 2 * kArcsPerHalf; i += 1024) lds_arcs[i] = arcs[(size_t)wg * 2 * kArcsPerHalf + i];
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const float* cur_a = (s & 1) ? a1 : a0; float* nxt_a = (s & 1) ? a0 : a1;
    const float* cur_b = (s & 1) ? b1 : b0; float* nxt_b = (s & 1) ? b0 : b1;
    if (MODE & 1) {
      // partial sums of the previous frame (every workgroup re-reduces all of them)
      float ps = 0.f;
      for (int i = ht; i < (int)gridDim.x; i += 512) ps += parts[(size_t)(s & 1) * 2 * gridDim.x + half * gridDim.x + i];
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      const int w = ht >> 6, lane = ht & 63;
      float4 g[kK], x[kK]; float pr[kK];
#pragma unroll
      for (int j = 0; j < kK; ++j) {
        const int2 r = lds_arcs[half * kArcsPerHalf + (w * kK + j) * 64 + lane];
        pr[j] = __int_as_float(r.y);
        if (half == 0) { g[j] = *reinterpret_cast<const float4*>(cur_a + (size_t)r.x * 4); x[j] = make_float4(1.f, 1.f, 1.f, 1.f); }
        else { g[j] = *reinterpret_cast<const float4*>(cur_b + (size_t)r.x * 8); x[j] = *reinterpret_cast<const float4*>(cur_b + (size_t)r.x * 8 + 4); }
      }
      for (int i = ht; i < kRows * 4; i += 512) acc[half][i] = 0.f;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < kK; ++j) {
        sum.x += g[j].x * x[j].x * pr[j]; sum.y += g[j].y * x[j].y * pr[j]; sum.z += g[j].z * x[j].z * pr[j]; sum.w += g[j].w * x[j].w * pr[j];
        if (j == 3 || j == 7) {
          const int row = (ht * 2 + (j >> 2)) & (kRows - 1);
          atomicAdd(&acc[half][row * 4 + 0], sum.x); atomicAdd(&acc[half][row * 4 + 1], sum.y);
          atomicAdd(&acc[half][row * 4 + 2], sum.z); atomicAdd(&acc[half][row * 4 + 3], sum.w);
          sum = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      __syncthreads();
      const int rows_per_wg = S / gridDim.x;
      if (ht < rows_per_wg) {
        const int row = wg * rows_per_wg + ht;
        float4 v = *reinterpret_cast<float4*>(&acc[half][(ht & (kRows - 1)) * 4]);
        const float sc = 1.0f / (1.0f + fabsf(ps) + fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w));
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        if (half == 0) *reinterpret_cast<float4*>(nxt_a + (size_t)row * 4) = v;
        else *reinterpret_cast<float4*>(nxt_b + (size_t)row * 8) = v;
      }
      if (ht == 0) parts[(size_t)((s + 1) & 1) * 2 * gridDim.x + half * gridDim.x + wg] = acc[half][0];
    }
    if ((MODE >> 1) == 0) { if (!grid_barrier(ctr, (unsigned)(s + 1) * gridDim.x, fail)) return; }
    else if ((MODE >> 1) == 1) { if (!flag_barrier<false>(flags, (unsigned)(s + 1), fail)) return; }
    else if ((MODE >> 1) == 2) { if (!flag_barrier<true>(flags, (unsigned)(s + 1), fail)) return; }
    else __syncthreads();
  }
}

int main() {
  int dev_cus = 0;
  CK(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int nwg = dev_cus, S = 30000 / nwg * nwg, steps = 600;
  std::vector<int2> h((size_t)nwg * 2 * kArcsPerHalf);
  srand(1);
  for (auto& r : h) { r.x = rand() % S; float p = 0.001f * (rand() % 1000); r.y = *reinterpret_cast<int*>(&p); }
  int2* arcs; float *a0, *a1, *b0, *b1, *parts; unsigned *ctr, *fail;
  CK(hipMalloc(&arcs, h.size() * 8)); CK(hipMemcpy(arcs, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&a0, S * 16)); CK(hipMalloc(&a1, S * 16)); CK(hipMalloc(&b0, S * 32)); CK(hipMalloc(&b1, S * 32));
  CK(hipMemset(a0, 0, S * 16)); CK(hipMemset(a1, 0, S * 16)); CK(hipMemset(b0, 0, S * 32)); CK(hipMemset(b1, 0, S * 32));
  CK(hipMalloc(&parts, 4 * nwg * 4)); CK(hipMemset(parts, 0, 4 * nwg * 4));
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&fail, 4)); CK(hipMemset(fail, 0, 4));
  const size_t lds = 2 * kArcsPerHalf * 8;
  unsigned* flags; CK(hipMalloc(&flags, 4 * nwg));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 8; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(ctr, 0, 4)); CK(hipMemset(flags, 0, 4 * nwg));
      CK(hipEventRecord(e0, 0));
#define L(M) case M: hipLaunchKernelGGL(frames<M>, dim3(nwg), dim3(1024), lds, 0, arcs, a0, a1, b0, b1, S, steps, ctr, fail, parts, flags); break;
      switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) }
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
    const char* bn[4] = {"counter barrier + fences", "flag barrier, no fences", "flag barrier + fences", "no grid barrier"};
    printf("%d workgroups x 1024 threads, %-28s %-26s %6.2f us/frame (barrier timeouts %u)\n", nwg,
           (mode & 1) ? "gather+reduce+write," : "barrier only,", bn[mode >> 1], 1e3 * ms / steps, f);
  }
  return 0;
}

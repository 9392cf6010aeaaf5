// Micro-benchmark (development aid): cost of one all-gather step among the workgroups of ONE XCD of a persistent
// kernel on MI355X -- the exchange a persistent LSTM direction confined to one XCD (32 CUs, W_hh slice of 128 KB
// resident in each CU's LDS) would pay per time step.  Workgroups on the same XCD share an L2, so the exchange
// can use L1-bypassing loads (sc0) that hit in L2 instead of device-scope (sc1) traffic.
// Also reports the blockIdx -> XCD mapping (hardware XCC_ID register) the design relies on.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned long long u64;

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ u64 load_sc0(const u64* p) {
  u64 v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// MODE 0: agent-scope atomics (sc1), MODE 1: plain stores + sc0 loads (L2 of the XCD is the meeting point)
template <int PER_WG, int MODE>
__global__ void __launch_bounds__(256) k(u64* buf, int steps, unsigned* fail, float* sink, int* xcd_of_wg, int use_xcd,
                                         unsigned* reg /* [9]: members per XCD, [8] = total registered */) {
  extern __shared__ float lds[];
  __shared__ int s_slot, s_group;
  const int wg = blockIdx.x, tid = threadIdx.x;
  const unsigned xcd = xcc_id();
  // every workgroup registers with its XCD; the members of an XCD are numbered in arrival order
  if (tid == 0) {
    xcd_of_wg[wg] = (int)xcd;
    s_slot = (int)__hip_atomic_fetch_add(&reg[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&reg[8], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&reg[8], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
      if (++spins > 20000000u) { atomicAdd(fail, 1000000u); break; }
    }
    s_group = (int)__hip_atomic_load(&reg[xcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if ((int)xcd != use_xcd) return;                 // only the workgroups of one XCD take part
  const int slot_in_xcd = s_slot, group = s_group;
  const int total = group * PER_WG;
  float acc = 0.f;
  for (int s = 1; s <= steps; ++s) {
    if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // somebody timed out
    u64* slot = buf + (size_t)(s & 1) * 64 * PER_WG;
    if (tid < PER_WG) {
      const u64 v = ((u64)s << 32) | (unsigned)(slot_in_xcd * PER_WG + tid + s);
      if (MODE == 0) __hip_atomic_store(slot + (size_t)slot_in_xcd * PER_WG + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else { slot[(size_t)slot_in_xcd * PER_WG + tid] = v; }
    }
    unsigned spins = 0;
    for (int i = tid; i < total; i += 256) {
      u64 v;
      do {
        v = MODE == 0 ? __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : load_sc0(slot + i);
        if (++spins > 200000u) { atomicAdd(fail, 1u); break; }
      } while ((unsigned)(v >> 32) != (unsigned)s);
      lds[i & 4095] = __uint_as_float((unsigned)v);
    }
    __syncthreads();
    acc += lds[(tid * 7) & 4095];
    __syncthreads();
  }
  if (acc == 1.2345f) sink[0] = acc;
}

// Sentinel exchange: every step has its own slot array pre-filled with a NaN pattern no result can take; a
// workgroup publishes VALS floats with agent-scope stores and polls the whole array 16 bytes at a time until no
// sentinel is left (no tags: 4 bytes per value).
constexpr unsigned kSentinel = 0x7fc0dead;
template <int VALS>
__global__ void __launch_bounds__(256) ks(unsigned* buf, int steps, unsigned* fail, float* sink, int use_xcd, unsigned* reg) {
  extern __shared__ float lds[];
  __shared__ int s_slot, s_group;
  const int tid = threadIdx.x;
  const unsigned xcd = xcc_id();
  if (tid == 0) {
    s_slot = (int)__hip_atomic_fetch_add(&reg[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&reg[8], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&reg[8], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x)
      if (++spins > 20000000u) { atomicAdd(fail, 1000000u); break; }
    s_group = (int)__hip_atomic_load(&reg[xcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if ((int)xcd != use_xcd) return;
  const int me = s_slot, group = s_group, total4 = group * VALS / 4;
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    unsigned* slot = buf + (size_t)s * 32 * VALS;
    if (tid < VALS)
      __hip_atomic_store(slot + (size_t)me * VALS + tid, __float_as_uint(0.001f * (me + tid + s)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    for (int i = tid; i < total4; i += 256) {
      const u64* p2 = reinterpret_cast<const u64*>(slot) + 2 * i;
      u64 lo, hi;
      do {
        lo = __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hi = __hip_atomic_load(p2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spins > 200000u) { atomicAdd(fail, 1u); break; }
      } while ((unsigned)lo == kSentinel || (unsigned)(lo >> 32) == kSentinel || (unsigned)hi == kSentinel || (unsigned)(hi >> 32) == kSentinel);
      *reinterpret_cast<float4*>(&lds[(i * 4) & 8191]) = make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                                                                    __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
    }
    __syncthreads();
    acc += lds[(tid * 7) & 8191];
    __syncthreads();
  }
  if (acc == 1.2345f) sink[0] = acc;
}

typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4u load16_sc0(const unsigned* p) {
  v4u v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// same exchange through the XCD's L2 only: plain stores (L1 is write-through) and L1-bypassing sc0 loads
template <int VALS>
__global__ void __launch_bounds__(256) ks0(unsigned* buf, int steps, unsigned* fail, float* sink, int use_xcd, unsigned* reg) {
  extern __shared__ float lds[];
  __shared__ int s_slot, s_group;
  const int tid = threadIdx.x;
  const unsigned xcd = xcc_id();
  if (tid == 0) {
    s_slot = (int)__hip_atomic_fetch_add(&reg[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&reg[8], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&reg[8], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x)
      if (++spins > 20000000u) { atomicAdd(fail, 1000000u); break; }
    s_group = (int)__hip_atomic_load(&reg[xcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if ((int)xcd != use_xcd) return;
  const int me = s_slot, group = s_group, total4 = group * VALS / 4;
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    unsigned* slot = buf + (size_t)s * 32 * VALS;
    if (tid < VALS) slot[(size_t)me * VALS + tid] = __float_as_uint(0.001f * (me + tid + s));
    unsigned spins = 0;
    for (int i = tid; i < total4; i += 256) {
      v4u v;
      do {
        v = load16_sc0(slot + 4 * i);
        if (++spins > 2000000u) { atomicAdd(fail, 1u); break; }
      } while (v.x == kSentinel || v.y == kSentinel || v.z == kSentinel || v.w == kSentinel);
      *reinterpret_cast<float4*>(&lds[(i * 4) & 8191]) = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    __syncthreads();
    acc += lds[(tid * 7) & 8191];
    __syncthreads();
  }
  if (acc == 1.2345f) sink[0] = acc;
}

template <int VALS>
int run_sentinel0(int steps) {
  const int nwg = 256;
  unsigned *buf, *fail, *reg; float* sink;
  const size_t words = (size_t)steps * 32 * VALS;
  CK(hipMalloc(&buf, 4 * words)); CK(hipMalloc(&reg, 36)); CK(hipMalloc(&fail, 4)); CK(hipMemset(fail, 0, 4)); CK(hipMalloc(&sink, 4));
  std::vector<unsigned> fill(words, kSentinel);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ks0<VALS>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemcpy(buf, fill.data(), 4 * words, hipMemcpyHostToDevice));
    CK(hipMemset(reg, 0, 36));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((ks0<VALS>), dim3(nwg), dim3(256), 100 * 1024, 0, buf, steps, fail, sink, 0, reg);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
  printf("one XCD, 32 workgroups, %3d floats/wg (%5d B gathered/step), sentinel slots, plain stores + sc0 16-byte polls: %6.2f us/step (timeouts %u)\n",
         VALS, 32 * VALS * 4, 1e3 * ms / steps, f);
  return 0;
}

template <int VALS>
int run_sentinel(int steps) {
  const int nwg = 256;
  unsigned *buf, *fail, *reg; float* sink;
  const size_t words = (size_t)steps * 32 * VALS;
  CK(hipMalloc(&buf, 4 * words)); CK(hipMalloc(&reg, 36)); CK(hipMalloc(&fail, 4)); CK(hipMemset(fail, 0, 4)); CK(hipMalloc(&sink, 4));
  std::vector<unsigned> fill(words, kSentinel);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ks<VALS>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemcpy(buf, fill.data(), 4 * words, hipMemcpyHostToDevice));
    CK(hipMemset(reg, 0, 36));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((ks<VALS>), dim3(nwg), dim3(256), 100 * 1024, 0, buf, steps, fail, sink, 0, reg);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
  printf("one XCD, 32 workgroups, %3d floats/wg (%5d B gathered/step), sentinel slots, 16-byte polls: %6.2f us/step (timeouts %u)\n",
         VALS, 32 * VALS * 4, 1e3 * ms / steps, f);
  return 0;
}

template <int PER_WG, int MODE>
int run(int steps) {
  const int nwg = 256;
  u64* buf; unsigned* fail; float* sink; int* xcd; unsigned* reg;
  CK(hipMalloc(&reg, 4 * 9));
  CK(hipMalloc(&buf, sizeof(u64) * 2 * nwg * PER_WG));
  CK(hipMalloc(&fail, 4)); CK(hipMemset(fail, 0, 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&xcd, 4 * nwg));
  const size_t lds = 100 * 1024;                  // one workgroup per CU
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<PER_WG, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(buf, 0, sizeof(u64) * 2 * nwg * PER_WG));
    CK(hipMemset(reg, 0, 4 * 9));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<PER_WG, MODE>), dim3(nwg), dim3(256), lds, 0, buf, steps, fail, sink, xcd, 0, reg);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
  std::vector<int> h(nwg); CK(hipMemcpy(h.data(), xcd, 4 * nwg, hipMemcpyDeviceToHost));
  int rr = 1; for (int i = 0; i < nwg; ++i) rr &= (h[i] == i % 8);
  unsigned hreg[9]; CK(hipMemcpy(hreg, reg, 36, hipMemcpyDeviceToHost));
  printf("one XCD, %u workgroups, %2d granules/wg (%5d B gathered/step), %s: %6.2f us/step (timeouts %u; wg->xcd round-robin: %s)\n",
         hreg[0], PER_WG, (int)hreg[0] * PER_WG * 8, MODE == 0 ? "agent-scope atomics (sc1)" : "plain store + sc0 loads    ", 1e3 * ms / steps, f,
         rr ? "yes" : "NO");
  if (!rr) { printf("  wg->xcd:"); for (int i = 0; i < 32; ++i) printf(" %d", h[i]); printf(" ...\n"); }
  return 0;
}

int main() {
  const int steps = 2000;
  run<32, 0>(steps);     // 16 units x 4 batch rows per workgroup, 2 values per granule
  run<64, 0>(steps);
  run<32, 1>(steps);
  run<64, 1>(steps);
  run_sentinel0<64>(steps);
  run_sentinel0<128>(steps);
  run_sentinel0<256>(steps);
  run_sentinel<64>(steps);     // forward: 16 units x 4 batch rows per workgroup -> 8 KB gathered
  run_sentinel<128>(steps);    // batch 8
  run_sentinel<256>(steps);    // backward: 4 gates x 16 units x 4 batch rows -> 32 KB gathered
  return 0;
}

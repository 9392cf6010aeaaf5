// Micro-benchmark (development aid): cost of one in-kernel all-gather step among the workgroups of a
// persistent kernel on MI355X, the exchange a persistent LSTM would need every time step.
// Each of NWG workgroups publishes PER_WG 8-byte {value, tag} granules (sc1 stores); every workgroup then
// sweeps the granules of its group (GROUP workgroups) with sc1 loads until all tags equal the step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned long long u64;

template <int PER_WG>
__global__ void __launch_bounds__(256) k(u64* buf, int group, int steps, unsigned* fail, float* sink) {
  __shared__ float lds[4096];
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int g0 = (wg / group) * group;           // first workgroup of my group
  const int total = group * PER_WG;               // granules to gather per step
  float acc = 0.f;
  for (int s = 1; s <= steps; ++s) {
    u64* slot = buf + (size_t)(s & 1) * gridDim.x * PER_WG;
    if (tid < PER_WG) {
      const u64 v = ((u64)s << 32) | (unsigned)(wg * PER_WG + tid + s);
      __hip_atomic_store(slot + (size_t)wg * PER_WG + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // sweep: thread i polls granules i, i+256, ...
    unsigned spins = 0;
    for (int i = tid; i < total; i += 256) {
      u64 v;
      do {
        v = __hip_atomic_load(slot + (size_t)g0 * PER_WG + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spins > 2000000u) { atomicAdd(fail, 1u); break; }
      } while ((unsigned)(v >> 32) != (unsigned)s);
      lds[i & 4095] = __uint_as_float((unsigned)v);
    }
    __syncthreads();
    acc += lds[(tid * 7) & 4095];
    __syncthreads();
  }
  if (acc == 1.2345f) sink[0] = acc;
}

template <int PER_WG>
int run(int nwg, int group, int steps) {
  u64* buf; unsigned* fail; float* sink;
  CK(hipMalloc(&buf, sizeof(u64) * 2 * nwg * PER_WG)); CK(hipMemset(buf, 0, sizeof(u64) * 2 * nwg * PER_WG));
  CK(hipMalloc(&fail, 4)); CK(hipMemset(fail, 0, 4)); CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(buf, 0, sizeof(u64) * 2 * nwg * PER_WG));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<PER_WG>, dim3(nwg), dim3(256), 0, 0, buf, group, steps, fail, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
  printf("nwg %3d group %3d granules/wg %3d (%5d B gathered/step): %6.2f us/step  (timeouts %u)\n", nwg, group, PER_WG,
         group * PER_WG * 8, 1e3 * ms / steps, f);
  CK(hipFree(buf)); CK(hipFree(fail)); CK(hipFree(sink));
  return 0;
}

int main() {
  const int steps = 2000;
  run<16>(256, 128, steps);   // forward LSTM: 2 directions x 128 WGs, 4 units x 4 batch = 16 values per WG
  run<16>(128, 64, steps);
  run<64>(64, 32, steps);     // 16 units per WG
  run<16>(64, 32, steps);
  run<64>(256, 128, steps);   // backward: 4 gates x 4 units x 4 batch = 64 values per WG
  run<16>(256, 256, steps);
  return 0;
}

// One-way latency of a flag between two workgroups on the SAME XCD (ping-pong), for the store / load flavours the
// persistent kernels could use.  hipcc --offload-arch=gfx950 -O3 flag_latency.hip -o flag_latency
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ unsigned xcc() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7; }
template <int MODE>
__global__ void pingpong(unsigned* flags, unsigned* who, int iters, long long* out) {
  // MODE 0: agent-scope store + agent-scope load; 1: atomic add + agent-scope load; 2: atomic add + atomic add(0) poll;
  // 3: agent store + atomic or(0) poll
  __shared__ int role;
  if (threadIdx.x == 0) {
    role = -1;
    if (xcc() == 0) role = (int)atomicAdd(who, 1u);
  }
  __syncthreads();
  if (role != 0 && role != 1) return;
  if (threadIdx.x != 0) return;
  unsigned* mine = flags + role * 64;        // separate cache lines
  unsigned* other = flags + (1 - role) * 64;
  const long long t0 = wall_clock64();
  for (int i = 1; i <= iters; ++i) {
    if (role == 0) {
      if (MODE == 0 || MODE == 3) __hip_atomic_store(mine, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    long long spin = 0;
    for (;;) {
      unsigned v;
      if (MODE == 0 || MODE == 1) v = __hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 2) v = __hip_atomic_fetch_add(other, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else v = __hip_atomic_fetch_or(other, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= (unsigned)i) break;
      if (++spin > 2000000) { out[2] = -1; return; }
    }
    if (role == 1) {
      if (MODE == 0 || MODE == 3) __hip_atomic_store(mine, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  out[role] = wall_clock64() - t0;
}
template <int MODE> void run(const char* name) {
  unsigned* flags; unsigned* who; long long* out;
  hipMalloc(&flags, 1024); hipMalloc(&who, 4); hipMalloc(&out, 32);
  hipMemset(flags, 0, 1024); hipMemset(who, 0, 4); hipMemset(out, 0, 32);
  const int iters = 2000;
  hipLaunchKernelGGL(pingpong<MODE>, dim3(64), dim3(64), 0, 0, flags, who, iters, out);
  hipDeviceSynchronize();
  long long h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
  printf("%-44s round trip %.0f ns  (one way %.0f ns)%s\n", name, 10.0 * h[0] / iters, 5.0 * h[0] / iters, h[2] ? "  TIMEOUT" : "");
  hipFree(flags); hipFree(who); hipFree(out);
}
int main() {
  run<0>("agent store + agent load");
  run<1>("atomic add + agent load");
  run<2>("atomic add + atomic add(0) poll");
  run<3>("agent store + atomic or(0) poll");
  return 0;
}

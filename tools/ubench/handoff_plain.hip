// Is "plain stores -> s_waitcnt vmcnt(0) -> workgroup barrier -> flag" a valid release between the CUs of ONE XCD?
// (The persistent kernels hand state vectors over that way inside an XCD; agent-scope stores are written through the L2 to
// the fabric on this multi-XCD part and cost 2x in the LSTM mailboxes.)  A team of 32 workgroups on one XCD, formed by
// XCC_ID like the kernels' teams; every iteration each workgroup rewrites its 1 KB slice of a 32 KB vector with the
// iteration's tag, waits for its stores, and publishes a flag; every workgroup then polls the 32 flags (L1-bypassing
// loads) and reads the WHOLE vector with L1-bypassing loads: a word that still carries an older tag is a stale read.
// Uneven load: odd workgroups stream 256 KB of unrelated memory before their stores in every third iteration.
//   hipcc --offload-arch=gfx950 -O3 handoff_plain.hip -o handoff_plain;  ./handoff_plain [iterations] [mode]
// mode 0: plain data stores + agent-scope flag; 1: agent-scope data stores (reference); 2: plain data AND plain flag.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ unsigned xcc() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7; }
__device__ unsigned ld_sc1(const unsigned* p) { unsigned v; asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
struct Ctl { unsigned arrive; unsigned stale; unsigned timeouts; unsigned pad; unsigned flags[2][32][16]; };
template <int MODE>
__global__ void __launch_bounds__(256) team(Ctl* ctl, unsigned* vec /* [2][32][256] */, const float* junk, float* sink, int iters) {
  __shared__ int s_rank;
  if (threadIdx.x == 0) s_rank = xcc() == 0 ? (int)atomicAdd(&ctl->arrive, 1u) : -1;
  __syncthreads();
  const int rank = s_rank;
  if (rank < 0 || rank >= 32) return;
  const int tid = threadIdx.x;
  unsigned stale = 0;
  float acc = 0.f;
  for (int it = 1; it <= iters; ++it) {
    unsigned* buf = vec + (size_t)(it & 1) * 32 * 256;
    if ((rank & 1) && it % 3 == 0)
      for (int k = 0; k < 256; ++k) acc += junk[((size_t)rank * 256 + k) * 256 + tid];
    if (MODE == 1) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(buf + rank * 256 + tid), "v"((unsigned)it) : "memory");
    else buf[rank * 256 + tid] = (unsigned)it;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (MODE == 2) ctl->flags[it & 1][rank][0] = (unsigned)it;
      else __hip_atomic_store(&ctl->flags[it & 1][rank][0], (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 32) {
      long long spin = 0;
      while (ld_sc1(&ctl->flags[it & 1][tid][0]) != (unsigned)it)
        if (++spin > 20000000) { atomicAdd(&ctl->timeouts, 1u); break; }
    }
    __syncthreads();
    for (int k = tid; k < 32 * 256; k += 256) stale += ld_sc1(buf + k) != (unsigned)it;
    __syncthreads();      // nobody rewrites this buffer (two iterations on) before everybody has read it ... see below
    // (a workgroup reaches iteration it + 2 only after all flags of it + 1, which every workgroup publishes after reading it)
  }
  if (stale) atomicAdd(&ctl->stale, stale);
  if (acc == 12345.f) sink[0] = acc;
}
template <int MODE> void run(const char* name, int iters) {
  Ctl* ctl; unsigned* vec; float* junk; float* sink;
  hipMalloc(&ctl, sizeof(Ctl)); hipMalloc(&vec, 2 * 32 * 256 * 4); hipMalloc(&junk, (size_t)32 * 256 * 256 * 4); hipMalloc(&sink, 4);
  hipMemset(ctl, 0, sizeof(Ctl)); hipMemset(vec, 0, 2 * 32 * 256 * 4); hipMemset(junk, 0, (size_t)32 * 256 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(team<MODE>, dim3(512), dim3(256), 0, 0, ctl, vec, junk, sink, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  Ctl h; hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
  printf("%-52s %d iterations, %.2f us each: stale words %u of %.3g read, poll time-outs %u, team size %u\n", name, iters, 1e3 * ms / iters,
         h.stale, (double)iters * 32 * 32 * 256, h.timeouts, h.arrive < 32 ? h.arrive : 32);
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200000;
  run<1>("agent-scope data stores + agent-scope flag", iters);
  run<0>("plain data stores + vmcnt(0) + agent-scope flag", iters);
  run<2>("plain data stores + vmcnt(0) + plain flag", iters);
  return 0;
}

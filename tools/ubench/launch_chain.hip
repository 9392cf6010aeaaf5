// Micro-benchmark (development aid, not part of the library): per-launch cost of a serial chain of
// small dependent kernels replayed from a hipGraph on MI355X, as a function of what the kernel does.
//   hipcc --offload-arch=gfx950 -O3 launch_chain.hip -o launch_chain && ./launch_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct P { const int* idx; const float* src; float* dst; int n; };
// mode bits: 1 = read params through a pointer (extra dependent level); 2 = dependent index load;
// 4 = dependent gather; 8 = write `wr` floats per thread; 16 = __syncthreads x4
__global__ void k(const P* pp, P pv, int mode, int wr, float* sink) {
  const P p = (mode & 1) ? *pp : pv;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  int j = i;
  if (mode & 2) j = p.idx[i % p.n];
  if (mode & 4) acc += p.src[(size_t)(j % p.n)];
  if (mode & 16) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
  for (int w = 0; w < wr; ++w) p.dst[(size_t)w * gridDim.x * blockDim.x + i] = acc + w;
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  const int n = 1 << 22;
  int* idx; float *src, *dst, *sink; P* dp;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&dst, (size_t)n * 4 * 8)); CK(hipMalloc(&sink, 4));
  CK(hipMalloc(&dp, sizeof(P)));
  std::vector<int> h(n); for (int i = 0; i < n; ++i) h[i] = (int)((1103515245u * i + 12345u) % n);
  CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemset(src, 0, n * 4));
  P pv{idx, src, dst, n};
  CK(hipMemcpy(dp, &pv, sizeof(P), hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreate(&s));
  struct Cfg { const char* name; int grid, block, mode, wr; size_t lds; };
  Cfg cfgs[] = {
    {"empty 256x256", 256, 256, 0, 0, 0},
    {"empty 256x512", 256, 512, 0, 0, 0},
    {"empty 256x1024", 256, 1024, 0, 0, 0},
    {"empty 256x512 lds128K", 256, 512, 0, 0, 128 * 1024},
    {"params via pointer", 256, 512, 1, 0, 0},
    {"+1 dependent load", 256, 512, 1 | 2, 0, 0},
    {"+2 dependent loads (gather)", 256, 512, 1 | 2 | 4, 0, 0},
    {"gather only (args by value)", 256, 512, 2 | 4, 0, 0},
    {"+4 barriers", 256, 512, 1 | 2 | 4 | 16, 0, 0},
    {"+write 1 float/thread (512KB)", 256, 512, 1 | 2 | 4 | 16, 1, 0},
    {"+write 4 floats/thread (2MB)", 256, 512, 1 | 2 | 4 | 16, 4, 0},
    {"write only 1 float/thread", 256, 512, 0, 1, 0},
    {"64 blocks x1024 gather+write", 64, 1024, 1 | 2 | 4, 1, 0},
  };
  for (const Cfg& c : cfgs) {
    if (c.lds) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int j = 0; j < 64; ++j) hipLaunchKernelGGL(k, dim3(c.grid), dim3(c.block), c.lds, s, dp, pv, c.mode, c.wr, sink);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    const int reps = 20;
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // eager for comparison
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 64 * 5; ++r) hipLaunchKernelGGL(k, dim3(c.grid), dim3(c.block), c.lds, s, dp, pv, c.mode, c.wr, sink);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
    printf("%-36s graph %6.2f us/launch   eager %6.2f us/launch\n", c.name, 1e3 * ms / (reps * 64), 1e3 * ms2 / (64 * 5));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}

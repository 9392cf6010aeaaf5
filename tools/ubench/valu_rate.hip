// Issue rate of v_fma_f32 / v_pk_fma_f32 / ds_read_b128 for ONE wave per SIMD against two (the persistent recurrences run
// one wave per SIMD): shader clocks per instruction.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, long long* clk, int iters) {
  __shared__ f4 lds[1024];
  lds[threadIdx.x] = f4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f2 a[8]; f2 w = {1.0001f, 0.9999f}, h = {0.5f, 0.25f};
  for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x, 1.f};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 64 packed FMAs, 8 independent chains
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(h));
    } else if (MODE == 1) {   // 128 plain FMAs, 16 independent chains
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i][0]) : "v"(w[0]), "v"(h[0]));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i][1]) : "v"(w[1]), "v"(h[1]));
        }
    } else if (MODE == 2) {   // 64 packed FMAs in ONE dependent chain
#pragma unroll
      for (int r = 0; r < 64; ++r) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(w), "v"(h));
    } else {                  // 16 ds_read_b128 + wait
      f4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = lds[(threadIdx.x & 15) * 9 + i];
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i & 7] += f2{v[i][0], v[i][3]};
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int threads, int per_iter) {
  float* out; long long* clk;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 8);
  const int iters = 1000;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, clk, iters);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, clk, iters);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  printf("%-52s %4d threads/CU: %.2f clocks per instruction (per wave)\n", name, threads, (double)h / iters / per_iter);
  hipFree(out); hipFree(clk);
}
int main() {
  run<0>("v_pk_fma_f32, 8 independent chains", 256, 64); run<0>("v_pk_fma_f32, 8 independent chains", 512, 64);
  run<1>("v_fma_f32, 16 independent chains", 256, 128);  run<1>("v_fma_f32, 16 independent chains", 512, 128);
  run<2>("v_pk_fma_f32, one dependent chain", 256, 64);  run<2>("v_pk_fma_f32, one dependent chain", 512, 64);
  run<3>("16 x ds_read_b128 + 16 packed adds", 256, 16); run<3>("16 x ds_read_b128 + 16 packed adds", 512, 16);
  return 0;
}

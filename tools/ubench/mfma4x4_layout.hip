// Development aid: operand / result layout of v_mfma_f32_4x4x1_16b_f32 on gfx950, checked against the mapping
// the small-batch LSTM step kernels assume:
//   16 independent blocks; lane l feeds block l/4:  A_blk[i = l%4][0] = a(l),  B_blk[0][j = l%4] = b(l);
//   result register r of lane l holds D_blk(l/4)[i = r][j = l%4].
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

int main() {
  float ha[64], hb[64], hd[256];
  for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3 * l; }
  float *a, *b, *d;
  CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&d, 1024));
  CK(hipMemcpy(a, ha, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
  CK(hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int blk = l / 4, j = l % 4, i = r;
      const float want = ha[blk * 4 + i] * hb[blk * 4 + j];
      if (hd[l * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, hd[l * 4 + r], want); ++bad; }
    }
  printf("mfma_f32_4x4x1f32 layout (block = lane/4, A row = lane%%4, B col = lane%%4, D[reg][lane%%4]): %s (%d mismatches)\n",
         bad ? "DIFFERENT" : "confirmed", bad);
  return 0;
}

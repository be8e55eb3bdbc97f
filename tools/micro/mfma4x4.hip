// Issue-rate probe for the rank-8 updates of attention: v_mfma_f32_4x4x1_16b_f32 (256 MACs per wave instruction) against
// v_pk_fma_f32 (128 MACs), alone and mixed with plain VALU work, w waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f2 = __attribute__((ext_vector_type(2))) float;
// MODE 0: 8 independent 4x4x1 MFMA accumulators; 1: 8 pk_fma chains; 2: MFMA + 4 v_fma per MFMA; 3: pk_fma + 4 v_fma per pk_fma
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x4 acc[8];
  f2 pk[8];
  float x[8];
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; pk[i] = f2{0.f, 0.f}; x[i] = threadIdx.x * 1e-3f + i; }
  const f2 aa = {a, a};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0 || MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[i], b, acc[i], 0, 0, 0);
      if (MODE == 1 || MODE == 3) pk[i] = __builtin_elementwise_fma(aa, f2{x[i], b}, pk[i]);
      if (MODE >= 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[(i + j) & 7] = fmaf(x[(i + j) & 7], a, b);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + pk[i][0] + pk[i][1] + x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char* names[4] = {"mfma4x4x1       ", "pk_fma          ", "mfma4x4x1 + 4fma", "pk_fma    + 4fma"};
  for (int mode = 0; mode < 4; ++mode)
    for (int w = 1; w <= 8; w *= 2) {
      const int grid = 256 * w;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double groups = (double)iters * 8 * w;      // (one MFMA or pk_fma [+ 4 fma]) groups per SIMD
      printf("%s waves/SIMD=%d: %.3f ms -> %.2f cycles per group per SIMD @2.4GHz\n", names[mode], w, ms, ms * 1e6 / groups * 2.4);
    }
  return 0;
}

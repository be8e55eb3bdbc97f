// packed-f32 VALU rate probe: v_pk_fma_f32 chains vs plain v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void kpk(float* out, int iters, float a, float b) {
  f2 x[8];
  for (int i = 0; i < 8; ++i) { x[i].x = threadIdx.x * 1e-3f + i; x[i].y = x[i].x + 0.5f; }
  const f2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
    int grid = 256 * wg_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kpk, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * 8 * wg_per_cu;
    printf("pk_fma waves/SIMD=%d: %.3f ms -> %.2f cyc@2.4GHz per wave-instr per SIMD (= %.2f per scalar fma)\n", wg_per_cu, ms,
           ms * 1e6 / instr_per_simd * 2.4, ms * 1e6 / instr_per_simd * 1.2);
  }
  return 0;
}

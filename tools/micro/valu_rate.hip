// VALU issue-rate probe: independent v_fma_f32 chains, w waves per SIMD.  Prints cycles per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], a, b);
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void kexp(float* out, int iters, float a, float b) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]) ;
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int which = 0; which < 2; ++which)
  for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
    int grid = 256 * wg_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f);
      else hipLaunchKernelGGL(kexp, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * 16 * wg_per_cu;          // each WG puts one wave on each SIMD
    printf("%s waves/SIMD=%d: %.3f ms -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", which ? "exp2" : "fma ", wg_per_cu, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  }
  return 0;
}

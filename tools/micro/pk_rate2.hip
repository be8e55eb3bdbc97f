// packed-f32 with a broadcast scalar operand (op_sel_hi:[0,1,1]) vs two plain v_fma_f32 on the same data flow.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f2 acc[8], v[4];
  float s[4];
  for (int i = 0; i < 8; ++i) { acc[i].x = threadIdx.x * 1e-3f + i; acc[i].y = acc[i].x + 0.5f; }
  for (int i = 0; i < 4; ++i) { v[i].x = a + i * 1e-5f; v[i].y = a - i * 1e-5f; s[i] = b + i * 1e-7f + threadIdx.x * 1e-9f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) { const f2 ss = {s[i & 3], s[i & 3]}; acc[i] = __builtin_elementwise_fma(ss, v[i & 3], acc[i]); }
      else { acc[i].x = fmaf(s[i & 3], v[i & 3].x, acc[i].x); acc[i].y = fmaf(s[i & 3], v[i & 3].y, acc[i].y); }
    }
  }
  float t = 0; for (int i = 0; i < 8; ++i) t += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int MODE> void run(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int w = 1; w <= 8; w *= 2) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s waves/SIMD=%d: %.2f cyc@2.4GHz per scalar fma\n", name, w, ms * 1e6 / ((double)iters * 16 * w) * 2.4);
  }
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * 8);
  run<0>("pk_fma op_sel-broadcast", out);
  run<1>("2 x v_fma              ", out);
  return 0;
}

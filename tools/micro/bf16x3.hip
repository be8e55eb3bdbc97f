// bf16x3.hip -- can the d = 8 score tile S^T = K Q^T run on the bf16 matrix path at fp32 accuracy?
// Each fp32 operand is split exactly into three bf16 pieces (x = x1 + x2 + x3); the six leading cross terms
// (q1k1, q1k2, q2k1, q1k3, q2k2, q3k1) are 48 products per (query, key) pair = three v_mfma_f32_32x32x16_bf16
// (the two lane halves carry two different terms) against four v_mfma_f32_32x32x2_f32.
// Prints the error of both forms against fp64 and the time of a loop of each with the same amount of vector work.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/bf16x3.hip -o gpurun_out/bf16x3 && gpurun_out/bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
using f16v = __attribute__((ext_vector_type(16))) float;
using f2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x; const float r = x - (float)a;
  b = (__bf16)r; const float r2 = r - (float)b;
  c = (__bf16)r2;
}
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// one wave: q, k are [32][8]; st[key][query]
__global__ void numerics(const float* q, const float* k, float* st_bf, float* st_f32) {
  const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
  __bf16 qp[3][8], kp[3][8];
  for (int d = 0; d < 8; ++d) { split3(q[l31 * 8 + d], qp[0][d], qp[1][d], qp[2][d]); split3(k[l31 * 8 + d], kp[0][d], kp[1][d], kp[2][d]); }
  // MFMA1: half0 (k1,q1) half1 (k2,q1); MFMA2: half0 (k1,q2) half1 (k3,q1); MFMA3: half0 (k2,q2) half1 (k1,q3)
  const int ka[3][2] = {{0, 1}, {0, 2}, {1, 0}}, qa[3][2] = {{0, 0}, {1, 0}, {1, 2}};
  f16v c = {0};
  for (int m = 0; m < 3; ++m) {
    bf8 a, b;
    for (int d = 0; d < 8; ++d) { a[d] = kp[ka[m][half]][d]; b[d] = qp[qa[m][half]][d]; }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  f16v c2 = {0};
  for (int s = 0; s < 4; ++s) c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(k[l31 * 8 + 2 * s + half], q[l31 * 8 + 2 * s + half], c2, 0, 0, 0);
  for (int r = 0; r < 16; ++r) { st_bf[acc_row(r, half) * 32 + l31] = c[r]; st_f32[acc_row(r, half) * 32 + l31] = c2[r]; }
}

template <int MODE>   // 0: fp32 MFMA x4; 1: bf16 MFMA x3; 2: none (vector work only)
__global__ __launch_bounds__(256, 2) void rate(const float* src, float* dst, int iters) {
  const int lane = threadIdx.x & 63;
  float a32[4], b32[4];
  bf8 a16[3], b16[3];
  for (int s = 0; s < 4; ++s) { a32[s] = src[lane + 64 * s]; b32[s] = src[256 + lane + 64 * s]; }
  for (int m = 0; m < 3; ++m)
    for (int d = 0; d < 8; ++d) { a16[m][d] = (__bf16)src[lane * 8 + d + m]; b16[m][d] = (__bf16)src[512 + lane * 8 + d + m]; }
  f2 acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  const f2 v[4] = {{src[0], src[1]}, {src[2], src[3]}, {src[4], src[5]}, {src[6], src[7]}};
  for (int it = 0; it < iters; ++it) {
    f16v c = {0};
    if (MODE == 0)
      for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a32[s], b32[s], c, 0, 0, 0);
    if (MODE == 1)
      for (int m = 0; m < 3; ++m) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16[m], b16[m], c, 0, 0, 0);
    if (MODE == 2)
      for (int r = 0; r < 16; ++r) c[r] = a32[r & 3] + (float)it;
    for (int r = 0; r < 16; ++r) {                     // the forward's vector work per score: exp + 4 packed FMAs
      const float p = __builtin_amdgcn_exp2f(c[r] - a32[0]);
      const f2 pp = {p, p};
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_elementwise_fma(pp, v[i], acc[i]);
    }
    a32[0] += 1e-9f;
  }
  float t = 0;
  for (int i = 0; i < 4; ++i) t += acc[i].x + acc[i].y;
  dst[blockIdx.x * 256 + threadIdx.x] = t;
}

int main() {
  std::vector<float> q(256), k(256);
  srand(1);
  for (auto& x : q) x = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  for (auto& x : k) x = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  float *dq, *dk, *d1, *d2;
  hipMalloc(&dq, 1024); hipMalloc(&dk, 1024); hipMalloc(&d1, 4096); hipMalloc(&d2, 4096);
  hipMemcpy(dq, q.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dk, k.data(), 1024, hipMemcpyHostToDevice);
  numerics<<<1, 64>>>(dq, dk, d1, d2);
  std::vector<float> s1(1024), s2(1024);
  hipMemcpy(s1.data(), d1, 4096, hipMemcpyDeviceToHost); hipMemcpy(s2.data(), d2, 4096, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0, mx = 0;
  for (int kk = 0; kk < 32; ++kk)
    for (int qq = 0; qq < 32; ++qq) {
      double ref = 0, mag = 0;
      for (int d = 0; d < 8; ++d) { ref += (double)k[kk * 8 + d] * q[qq * 8 + d]; mag += fabs((double)k[kk * 8 + d] * q[qq * 8 + d]); }
      e1 = fmax(e1, fabs(s1[kk * 32 + qq] - ref) / mag); e2 = fmax(e2, fabs(s2[kk * 32 + qq] - ref) / mag); mx = fmax(mx, mag);
    }
  printf("max |err| / sum|q k|: bf16x3 (6 terms) %.3e   fp32 mfma %.3e   (fp32 eps 5.96e-08)\n", e1, e2);
  float *src, *dst;
  hipMalloc(&src, 1 << 16); hipMemset(src, 0, 1 << 16); hipMalloc(&dst, 4096 * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000, grid = 2048;
  for (int mode = 0; mode < 3; ++mode) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (mode == 0) rate<0><<<grid, 256>>>(src, dst, iters);
      if (mode == 1) rate<1><<<grid, 256>>>(src, dst, iters);
      if (mode == 2) rate<2><<<grid, 256>>>(src, dst, iters);
      hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    }
    const double waves_per_simd = grid * 4.0 / 1024.0;
    printf("mode %d (%s): %.3f ms -> %.0f cycles per tile per SIMD at 2.4 GHz\n", mode,
           mode == 0 ? "4 x mfma_f32_32x32x2" : (mode == 1 ? "3 x mfma_f32_32x32x16_bf16" : "vector work only"), ms,
           ms * 1e-3 * 2.4e9 / (iters * waves_per_simd));
  }
  return 0;
}

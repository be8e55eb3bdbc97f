// Ablation probe for attn_fwd_mfma8: which of {MFMA chain, LDS row reads, global staging + barriers, VALU} bounds it.
// ABL bit 0: skip MFMAs (scores = cheap VALU)   bit 1: no LDS row reads (rows from registers)
// ABL bit 2: stage only the first tile (no global loads / barriers in the loop)   bit 3: skip the rank-8 FMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f2 = __attribute__((ext_vector_type(2))) float;
constexpr int kD = 8, kTK = 64, kWave = 64;
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, kWave); }
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ void load_row8(const float* __restrict__ p, f2 (&v)[4]) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = (f2){a.x, a.y}; v[1] = (f2){a.z, a.w}; v[2] = (f2){b.x, b.y}; v[3] = (f2){b.z, b.w};
}
__device__ __forceinline__ void axpy8(f2 (&acc)[4], float s, const f2 (&v)[4]) {
  const f2 ss = {s, s};
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = __builtin_elementwise_fma(ss, v[i], acc[i]);
}
template <int ABL>
__global__ __launch_bounds__(256, 2) void fwd(const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ lse,
                                              int heads, int L, float scale) {
  __shared__ __attribute__((aligned(16))) float Kd[kD * kTK];
  __shared__ __attribute__((aligned(16))) float Vr[kTK * kD];
  const int b = blockIdx.z, h = blockIdx.y, C = heads * kD;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const float* qp = qkv + ((long)b * 3 * C + h * kD) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const int q0 = blockIdx.x * 256 + wv * 64;
  float bq[2][4];
  for (int j = 0; j < 2; ++j)
    for (int s = 0; s < 4; ++s) bq[j][s] = qp[(long)(2 * s + half) * L + q0 + j * 32 + l31] * (scale * kLog2e);
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  f2 oa[2][4];
  for (int j = 0; j < 2; ++j) for (int i = 0; i < 4; ++i) oa[j][i] = (f2){0.f, 0.f};
  f2 vconst[4] = {{bq[0][0], bq[0][1]}, {bq[0][2], bq[0][3]}, {bq[1][0], bq[1][1]}, {bq[1][2], bq[1][3]}};
  for (int k0 = 0; k0 < L; k0 += kTK) {
    if (!(ABL & 4) || k0 == 0) {
      __syncthreads();
      for (int i = threadIdx.x; i < kD * kTK; i += 256) {
        const int j = i / kTK, rr = i % kTK;
        Kd[j * kTK + rr] = kp[(long)j * L + k0 + rr];
        Vr[rr * kD + j] = vp[(long)j * L + k0 + rr];
      }
      __syncthreads();
    }
#pragma unroll
    for (int kt = 0; kt < kTK / 32; ++kt) {
      float ak[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) ak[s] = Kd[(2 * s + half) * kTK + kt * 32 + l31];
      f32x16 sc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[j][r] = 0.f;
        if (ABL & 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[j][r] = ak[r & 3] * bq[j][(r >> 2) & 3];
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) sc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[s], bq[j][s], sc[j], 0, 0, 0);
        }
        float mx = sc[j][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[j][r]);
        mx = fmaxf(mx, xhalf(mx));
        const float mn = fmaxf(m[j], mx);
        const float alpha = __builtin_amdgcn_exp2f(m[j] - mn);
        l[j] *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i) oa[j][i] *= alpha;
        m[j] = mn;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f2 v[4];
        if (ABL & 2) { for (int i = 0; i < 4; ++i) v[i] = vconst[i]; }
        else load_row8(Vr + (kt * 32 + acc_row(r, half)) * kD, v);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float p = __builtin_amdgcn_exp2f(sc[j][r] - m[j]);
          l[j] += p;
          if (ABL & 8) oa[j][r & 3].x += p; else axpy8(oa[j], p, v);
        }
      }
    }
  }
  for (int j = 0; j < 2; ++j) {
    const float lt = l[j] + xhalf(l[j]);
    const float inv = 1.0f / lt;
    const int qi = q0 + j * 32 + l31;
    for (int d = 0; d < kD; ++d) {
      const float mine = oa[j][d >> 1][d & 1];
      const float t = mine + xhalf(mine);
      if (half == 0) o[((long)b * C + h * kD + d) * L + qi] = t * inv;
    }
    if (half == 0) lse[((long)b * heads + h) * L + qi] = m[j] + __builtin_amdgcn_logf(lt);
  }
}
template <int ABL> void run(const char* name, float* qkv, float* o, float* lse) {
  const int B = 256, heads = 4, L = 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(fwd<ABL>, dim3(L / 256, heads, B), dim3(256), 0, 0, qkv, o, lse, heads, L, 0.35f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%-40s %8.1f us\n", name, best * 1e3);
}
int main() {
  const size_t n = (size_t)256 * 96 * 1024;
  float *qkv, *o, *lse; hipMalloc(&qkv, n * 4); hipMalloc(&o, n * 4 / 3); hipMalloc(&lse, 256 * 4 * 1024 * 4);
  float* hq = (float*)malloc(n * 4); for (size_t i = 0; i < n; ++i) hq[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(qkv, hq, n * 4, hipMemcpyHostToDevice);
  run<0>("baseline", qkv, o, lse);
  run<1>("no MFMA", qkv, o, lse);
  run<2>("no LDS row reads", qkv, o, lse);
  run<4>("no global staging/barriers", qkv, o, lse);
  run<8>("no rank-8 FMAs", qkv, o, lse);
  run<3>("no MFMA, no LDS rows", qkv, o, lse);
  run<6>("no LDS rows, no staging", qkv, o, lse);
  run<7>("no MFMA, no LDS rows, no staging", qkv, o, lse);
  run<15>("only exp/softmax bookkeeping", qkv, o, lse);
  return 0;
}

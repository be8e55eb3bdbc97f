// attn.hip -- F10 self-attention core: O = softmax(Q K^T / sqrt(d)) V per (batch, head), flash-style
// (online softmax, the L x L scores never exist in memory).
//
// Layout: qkv (B, 3C, L) = the NCHW image of the in-projection, channel = {q,k,v}*C + head*d + j,
// token index contiguous; o (B, C, L); lse (B, heads, L).
//
// Shapes here are d in {2,4,8,16,32,64}, L in {16..4096}: with d = 8 a 32x32 or 16x16 MFMA tile would
// run 50-75 % empty and the exp / max / rescale work (VALU) is as large as the contractions, so
// the kernels are one-lane-per-row VALU kernels: the lane keeps its query (or key) row, the output
// accumulator and the running max / sum in registers; the other operand streams through LDS in
// tiles that every lane reads at the same address (LDS broadcast, conflict-free).
// Forward: lane = query.  Backward: one pass with lane = query (dQ), one with lane = key (dK, dV);
// delta = rowsum(dO * O) is recomputed per tile -- no atomics, bitwise reproducible.
#include "common.h"

namespace afd {

constexpr int kTile = 64;     // keys (or queries) staged per LDS tile
constexpr int kAttnBlock = 128;

template <int D>
__global__ __launch_bounds__(kAttnBlock) void attn_fwd_k(const float* __restrict__ qkv, float* __restrict__ o,
                                                         float* __restrict__ lse, int heads, int L, float scale) {
  __shared__ float Ks[D][kTile];
  __shared__ float Vs[D][kTile];
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = heads * D;
  const int qi = blockIdx.x * kAttnBlock + threadIdx.x;
  const bool live = qi < L;
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  float q[D], acc[D];
#pragma unroll
  for (int j = 0; j < D; ++j) { q[j] = live ? qp[(long)j * L + qi] * scale : 0.f; acc[j] = 0.f; }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < L; k0 += kTile) {
    __syncthreads();
    for (int i = threadIdx.x; i < D * kTile; i += kAttnBlock) {
      const int j = i / kTile, kk = i % kTile;
      const bool in = k0 + kk < L;
      Ks[j][kk] = in ? kp[(long)j * L + k0 + kk] : 0.f;
      Vs[j][kk] = in ? vp[(long)j * L + k0 + kk] : 0.f;
    }
    __syncthreads();
    const int nk = min(kTile, L - k0);
    for (int c0 = 0; c0 < nk; c0 += 8) {            // chunks of 8 keys: one rescale per chunk
      float s[8];
      float cm = m;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) a += q[j] * Ks[j][c0 + u];
        s[u] = (c0 + u < nk) ? a : -INFINITY;
        cm = fmaxf(cm, s[u]);
      }
      const float alpha = __expf(m - cm);            // m = -inf on the first chunk -> 0
      l *= alpha;
#pragma unroll
      for (int j = 0; j < D; ++j) acc[j] *= alpha;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float p = __expf(s[u] - cm);
        l += p;
#pragma unroll
        for (int j = 0; j < D; ++j) acc[j] += p * Vs[j][c0 + u];
      }
      m = cm;
    }
  }
  if (live) {
    const float inv = 1.0f / l;
    float* op = o + ((long)b * C + h * D) * L + qi;
#pragma unroll
    for (int j = 0; j < D; ++j) op[(long)j * L] = acc[j] * inv;
    lse[((long)b * heads + h) * L + qi] = m + __logf(l);
  }
}

// dQ: lane = query.  ds = p * (dp - delta) ; dq += ds * k * scale
template <int D>
__global__ __launch_bounds__(kAttnBlock) void attn_bwd_dq_k(const float* __restrict__ qkv, const float* __restrict__ o,
                                                            const float* __restrict__ d_o, const float* __restrict__ lse,
                                                            float* __restrict__ dqkv, int heads, int L, float scale) {
  __shared__ float Ks[D][kTile];
  __shared__ float Vs[D][kTile];
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = heads * D;
  const int qi = blockIdx.x * kAttnBlock + threadIdx.x;
  const bool live = qi < L;
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const long ooff = ((long)b * C + h * D) * L + qi;
  float q[D], go[D], dq[D];
  float delta = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    q[j] = live ? qp[(long)j * L + qi] * scale : 0.f;
    go[j] = live ? d_o[ooff + (long)j * L] : 0.f;
    delta += go[j] * (live ? o[ooff + (long)j * L] : 0.f);
    dq[j] = 0.f;
  }
  const float ls = live ? lse[((long)b * heads + h) * L + qi] : 0.f;
  for (int k0 = 0; k0 < L; k0 += kTile) {
    __syncthreads();
    for (int i = threadIdx.x; i < D * kTile; i += kAttnBlock) {
      const int j = i / kTile, kk = i % kTile;
      const bool in = k0 + kk < L;
      Ks[j][kk] = in ? kp[(long)j * L + k0 + kk] : 0.f;
      Vs[j][kk] = in ? vp[(long)j * L + k0 + kk] : 0.f;
    }
    __syncthreads();
    const int nk = min(kTile, L - k0);
    for (int u = 0; u < nk; ++u) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) { s += q[j] * Ks[j][u]; dp += go[j] * Vs[j][u]; }
      const float ds = __expf(s - ls) * (dp - delta);
#pragma unroll
      for (int j = 0; j < D; ++j) dq[j] += ds * Ks[j][u];
    }
  }
  if (live) {
    float* dqp = dqkv + ((long)b * 3 * C + h * D) * L + qi;
#pragma unroll
    for (int j = 0; j < D; ++j) dqp[(long)j * L] = dq[j] * scale;
  }
}

// dK, dV: lane = key.  Per query tile LDS holds Q*scale, dO, lse and delta.
template <int D>
__global__ __launch_bounds__(kAttnBlock) void attn_bwd_dkv_k(const float* __restrict__ qkv, const float* __restrict__ o,
                                                             const float* __restrict__ d_o, const float* __restrict__ lse,
                                                             float* __restrict__ dqkv, int heads, int L, float scale) {
  __shared__ float Qs[D][kTile];
  __shared__ float Gs[D][kTile];
  __shared__ float Ls[kTile];
  __shared__ float Ds[kTile];
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = heads * D;
  const int ki = blockIdx.x * kAttnBlock + threadIdx.x;
  const bool live = ki < L;
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const float* op = o + ((long)b * C + h * D) * L;
  const float* gp = d_o + ((long)b * C + h * D) * L;
  const float* lp = lse + ((long)b * heads + h) * L;
  float k[D], v[D], dk[D], dv[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    k[j] = live ? kp[(long)j * L + ki] : 0.f;
    v[j] = live ? vp[(long)j * L + ki] : 0.f;
    dk[j] = 0.f; dv[j] = 0.f;
  }
  for (int q0 = 0; q0 < L; q0 += kTile) {
    __syncthreads();
    for (int i = threadIdx.x; i < D * kTile; i += kAttnBlock) {
      const int j = i / kTile, qq = i % kTile;
      const bool in = q0 + qq < L;
      Qs[j][qq] = in ? qp[(long)j * L + q0 + qq] * scale : 0.f;
      Gs[j][qq] = in ? gp[(long)j * L + q0 + qq] : 0.f;
    }
    if (threadIdx.x < kTile) {
      const int qq = threadIdx.x;
      const bool in = q0 + qq < L;
      float dl = 0.f;
      if (in) {
#pragma unroll
        for (int j = 0; j < D; ++j) dl += gp[(long)j * L + q0 + qq] * op[(long)j * L + q0 + qq];
      }
      Ds[qq] = dl;
      Ls[qq] = in ? lp[q0 + qq] : INFINITY;          // exp(s - inf) = 0 for padded queries
    }
    __syncthreads();
    const int nq = min(kTile, L - q0);
    for (int u = 0; u < nq; ++u) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) { s += Qs[j][u] * k[j]; dp += Gs[j][u] * v[j]; }
      const float p = __expf(s - Ls[u]);
      const float ds = p * (dp - Ds[u]);
#pragma unroll
      for (int j = 0; j < D; ++j) { dv[j] += p * Gs[j][u]; dk[j] += ds * Qs[j][u]; }   // Qs already carries `scale`
    }
  }
  if (live) {
    float* dkp = dqkv + ((long)b * 3 * C + C + h * D) * L + ki;
    float* dvp = dkp + (long)C * L;
#pragma unroll
    for (int j = 0; j < D; ++j) { dkp[(long)j * L] = dk[j]; dvp[(long)j * L] = dv[j]; }
  }
}

}  // namespace afd
using namespace afd;

#define AFD_ATTN_DISPATCH(D_, KERNEL, ...)                                                     \
  switch (D_) {                                                                                \
    case 2:  hipLaunchKernelGGL(KERNEL<2>,  grid, dim3(kAttnBlock), 0, s, __VA_ARGS__); break; \
    case 4:  hipLaunchKernelGGL(KERNEL<4>,  grid, dim3(kAttnBlock), 0, s, __VA_ARGS__); break; \
    case 8:  hipLaunchKernelGGL(KERNEL<8>,  grid, dim3(kAttnBlock), 0, s, __VA_ARGS__); break; \
    case 16: hipLaunchKernelGGL(KERNEL<16>, grid, dim3(kAttnBlock), 0, s, __VA_ARGS__); break; \
    case 32: hipLaunchKernelGGL(KERNEL<32>, grid, dim3(kAttnBlock), 0, s, __VA_ARGS__); break; \
    default: hipLaunchKernelGGL(KERNEL<64>, grid, dim3(kAttnBlock), 0, s, __VA_ARGS__); break; \
  }

extern "C" {

int afd_attn_fwd(const float* qkv, float* o, float* lse, int B, int heads, int d, int L, afd_stream_t st) {
  AFD_REQUIRE(qkv && o && lse && B > 0 && heads > 0 && L > 0, "afd_attn_fwd: bad argument");
  AFD_REQUIRE(d == 2 || d == 4 || d == 8 || d == 16 || d == 32 || d == 64, "afd_attn_fwd: head dim %d not in {2,4,8,16,32,64}", d);
  AFD_REQUIRE(B <= 65535 && heads <= 65535, "afd_attn_fwd: grid too large");
  hipStream_t s = as_stream(st);
  const dim3 grid((L + kAttnBlock - 1) / kAttnBlock, heads, B);
  const float scale = 1.0f / sqrtf((float)d);
  AFD_ATTN_DISPATCH(d, attn_fwd_k, qkv, o, lse, heads, L, scale);
  return check_launch("afd_attn_fwd");
}

int afd_attn_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                 int B, int heads, int d, int L, afd_stream_t st) {
  AFD_REQUIRE(qkv && o && d_o && lse && dqkv && B > 0 && heads > 0 && L > 0, "afd_attn_bwd: bad argument");
  AFD_REQUIRE(d == 2 || d == 4 || d == 8 || d == 16 || d == 32 || d == 64, "afd_attn_bwd: head dim %d not in {2,4,8,16,32,64}", d);
  AFD_REQUIRE(B <= 65535 && heads <= 65535, "afd_attn_bwd: grid too large");
  hipStream_t s = as_stream(st);
  const dim3 grid((L + kAttnBlock - 1) / kAttnBlock, heads, B);
  const float scale = 1.0f / sqrtf((float)d);
  AFD_ATTN_DISPATCH(d, attn_bwd_dq_k, qkv, o, d_o, lse, dqkv, heads, L, scale);
  AFD_ATTN_DISPATCH(d, attn_bwd_dkv_k, qkv, o, d_o, lse, dqkv, heads, L, scale);
  return check_launch("afd_attn_bwd");
}

}  // extern "C"

// norm.hip -- F6 GroupNorm(1,C) (one group = the whole C*H*W sample) and the channel LayerNorm of
// the attention blocks, both on NCHW fp32.
//
// GroupNorm forward: one workgroup per sample, the sample stays in registers (float4 x NCH per
// thread) between the mean pass, the centred-variance pass and the apply, so HBM sees one read and
// one write (8 B/element; 4 B/element in stats-only mode, used when the apply is folded into the
// filtered-GELU kernel's load).  Samples too large for registers take a 3-pass loop (L2 re-reads).
// GroupNorm backward: (1) per-(b,c)-plane sums  A1 = sum dz, A2 = sum dz*xhat  (one wave per plane;
// these ARE the dgamma/dbeta partials), (2) elementwise dx from the per-sample contractions of A1/A2.
#include "common.h"

namespace afd {

// ------------------------------------------------------------------------------------------
// GroupNorm(1, C) forward
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float gn_epilogue(float xh, float g, float be, float r, int act, float e) {
  float z = xh * g + be + r;
  if (act == 1) z = gelu_erf(z);
  return z + e;
}

template <int NCH>   // float4 chunks per thread; n4 <= NCH * blockDim.x
__global__ __launch_bounds__(1024) void gn_fwd_reg(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ stats,
                                                   int C, int HW, float eps, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ res, int act,
                                                   const float* __restrict__ emb) {
  __shared__ float red[16];
  const long b = blockIdx.x;
  const long n = (long)C * HW;
  const int n4 = (int)(n >> 2);
  const float4* x4 = reinterpret_cast<const float4*>(x + b * n);
  float4 v[NCH];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int q = k * blockDim.x + threadIdx.x;
    v[k] = q < n4 ? x4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  const float mean = block_sum(s, red) / (float)n;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int q = k * blockDim.x + threadIdx.x;
    if (q < n4) {
      const float a = v[k].x - mean, bb = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
      s2 += (a * a + bb * bb) + (c * c + d * d);
    }
  }
  const float var = block_sum(s2, red) / (float)n;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) { stats[2 * b] = mean; stats[2 * b + 1] = rstd; }
  if (!y) return;
  float4* y4 = reinterpret_cast<float4*>(y + b * n);
  const float4* r4 = res ? reinterpret_cast<const float4*>(res + b * n) : nullptr;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int q = k * blockDim.x + threadIdx.x;
    if (q < n4) {
      const int c = (q << 2) / HW;                       // HW % 4 == 0 on this path: a chunk never straddles channels
      const float g = gamma[c], be = beta[c];
      const float e = emb ? emb[b * C + c] : 0.f;
      const float4 r = r4 ? r4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 o;
      o.x = gn_epilogue((v[k].x - mean) * rstd, g, be, r.x, act, e);
      o.y = gn_epilogue((v[k].y - mean) * rstd, g, be, r.y, act, e);
      o.z = gn_epilogue((v[k].z - mean) * rstd, g, be, r.z, act, e);
      o.w = gn_epilogue((v[k].w - mean) * rstd, g, be, r.w, act, e);
      y4[q] = o;
    }
  }
}

// any C, HW: three passes over global memory (the sample is L2-resident between passes)
__global__ __launch_bounds__(1024) void gn_fwd_loop(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ stats,
                                                    int C, int HW, float eps, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ res, int act,
                                                    const float* __restrict__ emb) {
  __shared__ float red[16];
  const long b = blockIdx.x;
  const long n = (long)C * HW;
  const float* xp = x + b * n;
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) s += xp[i];
  const float mean = block_sum(s, red) / (float)n;
  float s2 = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) { const float d = xp[i] - mean; s2 += d * d; }
  const float var = block_sum(s2, red) / (float)n;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) { stats[2 * b] = mean; stats[2 * b + 1] = rstd; }
  if (!y) return;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const int c = (int)(i / HW);
    y[b * n + i] = gn_epilogue((xp[i] - mean) * rstd, gamma[c], beta[c], res ? res[b * n + i] : 0.f, act,
                               emb ? emb[b * C + c] : 0.f);
  }
}

// ------------------------------------------------------------------------------------------
// GroupNorm(1, C) backward
// ------------------------------------------------------------------------------------------
// dz = dy * act'(z),  z = xhat*gamma + beta + res   (act' = 1 when act == 0)
__device__ __forceinline__ float gn_dz(float xh, float dyv, float g, float be, float r, int act) {
  if (act == 1) return dyv * gelu_erf_grad(xh * g + be + r);
  return dyv;
}

// one wave per (b, c) plane: part[b][0][c] = sum dz*xhat, part[b][1][c] = sum dz;  demb[b,c] = sum dy;  dres = dz
__global__ __launch_bounds__(256) void gn_bwd_plane(const float* __restrict__ x, const float* __restrict__ dy,
                                                    const float* __restrict__ stats, int C, int HW, long planes,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ res, int act,
                                                    float* __restrict__ dres, float* __restrict__ part, float* __restrict__ demb) {
  const long pl = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (pl >= planes) return;
  const long b = pl / C; const int c = pl % C;
  const float mean = stats[2 * b], rstd = stats[2 * b + 1];
  const float g = gamma[c], be = beta[c];
  const long base = pl * HW;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int i = lane; i < HW; i += 64) {
    const float xh = (x[base + i] - mean) * rstd;
    const float d = dy[base + i];
    const float dz = gn_dz(xh, d, g, be, res ? res[base + i] : 0.f, act);
    if (dres) dres[base + i] = dz;
    s1 += dz * xh; s2 += dz; s3 += d;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
  if (lane == 0) {                                  // partial layout (B, 2, C): [b][0][c] = sum dz*xhat, [b][1][c] = sum dz
    part[(2 * b) * C + c] = s1; part[(2 * b + 1) * C + c] = s2;
    if (demb) demb[pl] = s3;
  }
}

// dgamma[c] (+)= sum_b part[b][0][c], dbeta[c] (+)= sum_b part[b][1][c], folded into the tail of the kernel that runs
// after the partials exist: block `blk` of `nblk` takes channels blk, blk+nblk, ... (fixed order: deterministic).
// Saves one tiny latency-bound launch per normalisation site (44 per training step).
__device__ __forceinline__ void fold_param_grads(const float* __restrict__ part, int B, int C, float* __restrict__ dgamma,
                                                 float* __restrict__ dbeta, int accumulate, float* red, long blk, long nblk) {
  for (long c = blk; c < C; c += nblk) {
    float a = 0.f, b2 = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) { a += part[(2L * b) * C + c]; b2 += part[(2L * b + 1) * C + c]; }
    a = block_sum(a, red);
    b2 = block_sum(b2, red);
    if (threadIdx.x == 0) {
      dgamma[c] = accumulate ? dgamma[c] + a : a;
      dbeta[c] = accumulate ? dbeta[c] + b2 : b2;
    }
  }
}

// dx = rstd * (gamma*dz - m1 - xhat*m2), with m1 = mean_sample(gamma*dz), m2 = mean_sample(gamma*dz*xhat).
// grid = (slices of the sample, B): every block first contracts the (2, C) partials of its sample with gamma
// (C <= a few hundred values: cheaper than a separate launch), then applies its slice.
__global__ __launch_bounds__(256) void gn_bwd_apply(const float* __restrict__ x, const float* __restrict__ dy,
                                                    const float* __restrict__ stats, const float* __restrict__ part,
                                                    int C, int HW, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ res, int act,
                                                    float* __restrict__ dx, float* __restrict__ dgamma,
                                                    float* __restrict__ dbeta, int accumulate) {
  __shared__ float red[16];
  const long b = blockIdx.y;
  float a = 0.f, c2 = 0.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float g = gamma[c];
    a += g * part[(2 * b + 1) * C + c];
    c2 += g * part[(2 * b) * C + c];
  }
  const float inv_n = 1.0f / (float)((long)C * HW);
  const float m1 = block_sum(a, red) * inv_n;
  const float m2 = block_sum(c2, red) * inv_n;
  const float mean = stats[2 * b], rstd = stats[2 * b + 1];
  const long n = (long)C * HW;
  if ((HW & 3) == 0) {                               // 16 bytes per lane: four pixels of one channel
    const long n4 = n >> 2;
    const long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long lo = blockIdx.x * per, hi = min(n4, lo + per);
    const float4* x4 = reinterpret_cast<const float4*>(x + b * n);
    const float4* d4 = reinterpret_cast<const float4*>(dy + b * n);
    const float4* r4 = res ? reinterpret_cast<const float4*>(res + b * n) : nullptr;
    float4* o4 = reinterpret_cast<float4*>(dx + b * n);
    const int hw4 = HW >> 2;
    for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
      const int c = (int)(i / hw4);
      const float g = gamma[c], be = beta[c];
      const float4 xv = x4[i], dv = d4[i];
      const float4 rv = r4 ? r4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 o;
      { const float xh = (xv.x - mean) * rstd; o.x = rstd * (g * gn_dz(xh, dv.x, g, be, rv.x, act) - m1 - xh * m2); }
      { const float xh = (xv.y - mean) * rstd; o.y = rstd * (g * gn_dz(xh, dv.y, g, be, rv.y, act) - m1 - xh * m2); }
      { const float xh = (xv.z - mean) * rstd; o.z = rstd * (g * gn_dz(xh, dv.z, g, be, rv.z, act) - m1 - xh * m2); }
      { const float xh = (xv.w - mean) * rstd; o.w = rstd * (g * gn_dz(xh, dv.w, g, be, rv.w, act) - m1 - xh * m2); }
      o4[i] = o;
    }
  } else {
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long lo = blockIdx.x * per, hi = min(n, lo + per);
    for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
      const int c = (int)(i / HW);
      const long gi = b * n + i;
      const float g = gamma[c];
      const float xh = (x[gi] - mean) * rstd;
      const float dz = gn_dz(xh, dy[gi], g, beta[c], res ? res[gi] : 0.f, act);
      dx[gi] = rstd * (g * dz - m1 - xh * m2);
    }
  }
  if (dgamma) fold_param_grads(part, gridDim.y, C, dgamma, dbeta, accumulate, red, blockIdx.y * gridDim.x + blockIdx.x, (long)gridDim.x * gridDim.y);
}

// Sample-resident backward (round 3): one workgroup per sample keeps xhat and dz of the whole sample in registers (float4 x
// NCH per thread, as gn_fwd_reg), so x / dy are read ONCE (12 B per element instead of the 20 of the plane pass + apply pass)
// and one launch does the job of two.  The per-channel sums (= the dgamma / dbeta partials, part (B, 2, C), and demb) are
// segmented reductions over the HW / 4 consecutive threads that hold a channel's chunks: butterflies inside a wave, wave
// sums through LDS when a channel spans several waves.  Needs HW % 4 == 0, HW / 4 a power of two, n / 4 <= NCH * blockDim.
template <int NCH>
__global__ __launch_bounds__(1024) void gn_bwd_reg(const float* __restrict__ x, const float* __restrict__ dy,
                                                   const float* __restrict__ stats, int C, int HW,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ res, int act, float* __restrict__ dx,
                                                   float* __restrict__ dres, float* __restrict__ part, float* __restrict__ demb) {
  __shared__ float red[16];
  __shared__ float seg[NCH][16][3];                   // wave sums of (dz xhat, dz, dy), chunk row k, when a channel spans waves
  const long b = blockIdx.x;
  const long n = (long)C * HW;
  const int n4 = (int)(n >> 2), T = blockDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int G = HW >> 2;                              // threads per channel in a chunk row
  const float mean = stats[2 * b], rstd = stats[2 * b + 1];
  const float4* x4 = reinterpret_cast<const float4*>(x + b * n);
  const float4* d4 = reinterpret_cast<const float4*>(dy + b * n);
  const float4* r4 = (res && act == 1) ? reinterpret_cast<const float4*>(res + b * n) : nullptr;
  float4* z4 = dres ? reinterpret_cast<float4*>(dres + b * n) : nullptr;
  float4 xh[NCH], dz[NCH];
  float gk[NCH];
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int q = k * T + tid;
    const bool ok = q < n4;
    const int c = ok ? (q << 2) / HW : 0;
    const float g = gamma[c], be = beta[c];
    gk[k] = g;
    const float4 xv = ok ? x4[q] : make_float4(mean, mean, mean, mean);
    const float4 dv = ok ? d4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 rv = (ok && r4) ? r4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    xh[k] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
    dz[k] = make_float4(gn_dz(xh[k].x, dv.x, g, be, rv.x, act), gn_dz(xh[k].y, dv.y, g, be, rv.y, act),
                        gn_dz(xh[k].z, dv.z, g, be, rv.z, act), gn_dz(xh[k].w, dv.w, g, be, rv.w, act));
    if (ok && z4) z4[q] = dz[k];
    float a = (dz[k].x * xh[k].x + dz[k].y * xh[k].y) + (dz[k].z * xh[k].z + dz[k].w * xh[k].w);
    float s1 = (dz[k].x + dz[k].y) + (dz[k].z + dz[k].w);
    float s3 = (dv.x + dv.y) + (dv.z + dv.w);
    sa = fmaf(g, s1, sa); sb = fmaf(g, a, sb);
    // the channel's sums over its G consecutive threads
    const int gw = G < 64 ? G : 64;
    for (int o = 1; o < gw; o <<= 1) { a += __shfl_xor(a, o, kWave); s1 += __shfl_xor(s1, o, kWave); s3 += __shfl_xor(s3, o, kWave); }
    if (G <= 64) {
      if (ok && (tid & (G - 1)) == 0) {
        part[(2 * b) * C + c] = a; part[(2 * b + 1) * C + c] = s1;
        if (demb) demb[b * C + c] = s3;
      }
    } else if (lane == 0) { seg[k][wv][0] = a; seg[k][wv][1] = s1; seg[k][wv][2] = s3; }
  }
  const float inv_n = 1.0f / (float)n;
  const float m1 = block_sum(sa, red) * inv_n;         // (its barriers also publish seg)
  const float m2 = block_sum(sb, red) * inv_n;
  if (G > 64) {                                        // one thread per (chunk row, channel): add the channel's wave sums in wave order
    const int wpc = G >> 6, cpr = T / G;               // waves per channel, channels per chunk row
    for (int i = tid; i < NCH * cpr; i += T) {
      const int k = i / cpr, j = i - k * cpr;
      const int q = k * T + j * G;
      if (q < n4) {
        float a = 0.f, s1 = 0.f, s3 = 0.f;
        for (int w = 0; w < wpc; ++w) { a += seg[k][j * wpc + w][0]; s1 += seg[k][j * wpc + w][1]; s3 += seg[k][j * wpc + w][2]; }
        const int c = (q << 2) / HW;
        part[(2 * b) * C + c] = a; part[(2 * b + 1) * C + c] = s1;
        if (demb) demb[b * C + c] = s3;
      }
    }
  }
  float4* o4 = reinterpret_cast<float4*>(dx + b * n);
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int q = k * T + tid;
    if (q < n4) {
      const float g = gk[k];
      o4[q] = make_float4(rstd * (g * dz[k].x - m1 - xh[k].x * m2), rstd * (g * dz[k].y - m1 - xh[k].y * m2),
                          rstd * (g * dz[k].z - m1 - xh[k].z * m2), rstd * (g * dz[k].w - m1 - xh[k].w * m2));
    }
  }
}

// ------------------------------------------------------------------------------------------
// LayerNorm over C for every pixel of an NCHW tensor (tokens = pixels)
// ------------------------------------------------------------------------------------------
// generic-C fallback (three strided passes, L2 re-reads)
__global__ __launch_bounds__(256) void ln_c_fwd(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ stats,
                                                int C, int HW, long pixels, float eps,
                                                const float* __restrict__ gamma, const float* __restrict__ beta) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pixels) return;
  const long b = p / HW; const int l = p % HW;
  const float* xp = x + b * (long)C * HW + l;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += xp[(long)c * HW];
  const float mean = s / (float)C;
  float s2 = 0.f;
  for (int c = 0; c < C; ++c) { const float d = xp[(long)c * HW] - mean; s2 += d * d; }
  const float rstd = 1.0f / sqrtf(s2 / (float)C + eps);
  stats[2 * p] = mean; stats[2 * p + 1] = rstd;
  float* yp = y + b * (long)C * HW + l;
  for (int c = 0; c < C; ++c) yp[(long)c * HW] = (xp[(long)c * HW] - mean) * rstd * gamma[c] + beta[c];
}

// C known at compile time: the pixel's channel vector lives in registers -> one read, one write
template <int C>
__global__ __launch_bounds__(256) void ln_c_fwd_reg(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ stats,
                                                    int HW, long pixels, float eps,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pixels) return;
  const long b = p / HW; const int l = p % HW;
  const long off = b * (long)C * HW + l;
  float v[C];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { v[c] = x[off + (long)c * HW]; s += v[c]; }
  const float mean = s / (float)C;
  float s2 = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { const float d = v[c] - mean; s2 += d * d; }
  const float rstd = 1.0f / sqrtf(s2 / (float)C + eps);
  stats[2 * p] = mean; stats[2 * p + 1] = rstd;
#pragma unroll
  for (int c = 0; c < C; ++c) y[off + (long)c * HW] = (v[c] - mean) * rstd * gamma[c] + beta[c];
}

// channel-split forward (see ln_c_bwd_dx_split): PX pixels x SPLIT channel groups per workgroup, the pixel's mean and
// centred variance meet in LDS (two exchanges), the channel slice stays in registers
template <int C, int SPLIT>
__global__ __launch_bounds__(256) void ln_c_fwd_split(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ stats,
                                                      int HW, long pixels, float eps,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta) {
  constexpr int PX = 256 / SPLIT, CPT = C / SPLIT;
  __shared__ float ps[2][SPLIT][PX];
  const int pi = threadIdx.x % PX, g = threadIdx.x / PX;
  const long p = (long)blockIdx.x * PX + pi;
  const bool live = p < pixels;
  float v[CPT];
  float s = 0.f;
  long off = 0;
  if (live) {
    const long b = p / HW; const int l = p % HW;
    off = b * (long)C * HW + l;
#pragma unroll
    for (int i = 0; i < CPT; ++i) { v[i] = x[off + (long)(g + i * SPLIT) * HW]; s += v[i]; }
  }
  ps[0][g][pi] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int k = 0; k < SPLIT; ++k) mean += ps[0][k][pi];
  mean /= (float)C;
  float s2 = 0.f;
  if (live) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) { const float d = v[i] - mean; s2 += d * d; }
  }
  ps[1][g][pi] = s2;
  __syncthreads();
  if (!live) return;
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < SPLIT; ++k) var += ps[1][k][pi];
  const float rstd = 1.0f / sqrtf(var / (float)C + eps);
  if (g == 0) { stats[2 * p] = mean; stats[2 * p + 1] = rstd; }
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + i * SPLIT;
    y[off + (long)c * HW] = (v[i] - mean) * rstd * gamma[c] + beta[c];
  }
}

__global__ __launch_bounds__(256) void ln_c_bwd_dx(const float* __restrict__ x, const float* __restrict__ dy,
                                                   const float* __restrict__ stats, int C, int HW, long pixels,
                                                   const float* __restrict__ gamma, float* __restrict__ dx,
                                                   const float* __restrict__ part, int B, float* __restrict__ dgamma,
                                                   float* __restrict__ dbeta, int accumulate, const float* __restrict__ addp) {
  __shared__ float red[16];
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < pixels) {
    const long b = p / HW; const int l = p % HW;
    const long off = b * (long)C * HW + l;
    const float mean = stats[2 * p], rstd = stats[2 * p + 1];
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float g = gamma[c] * dy[off + (long)c * HW];
      s1 += g; s2 += g * ((x[off + (long)c * HW] - mean) * rstd);
    }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C;
    for (int c = 0; c < C; ++c) {
      const float xh = (x[off + (long)c * HW] - mean) * rstd;
      dx[off + (long)c * HW] = rstd * (gamma[c] * dy[off + (long)c * HW] - m1 - xh * m2) + (addp ? addp[off + (long)c * HW] : 0.f);
    }
  }
  if (dgamma) fold_param_grads(part, B, C, dgamma, dbeta, accumulate, red, blockIdx.x, gridDim.x);
}

// register form: gamma*dy is kept in registers, x is read twice (second read hits L2)
template <int C>
__global__ __launch_bounds__(256) void ln_c_bwd_dx_reg(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ stats, int HW, long pixels,
                                                       const float* __restrict__ gamma, float* __restrict__ dx,
                                                       const float* __restrict__ part, int B, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, int accumulate, const float* __restrict__ addp) {
  __shared__ float red[16];
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < pixels) {
    const long b = p / HW; const int l = p % HW;
    const long off = b * (long)C * HW + l;
    const float mean = stats[2 * p], rstd = stats[2 * p + 1];
    float g[C];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      g[c] = gamma[c] * dy[off + (long)c * HW];
      s1 += g[c]; s2 += g[c] * ((x[off + (long)c * HW] - mean) * rstd);
    }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float xh = (x[off + (long)c * HW] - mean) * rstd;
      dx[off + (long)c * HW] = rstd * (g[c] - m1 - xh * m2) + (addp ? addp[off + (long)c * HW] : 0.f);
    }
  }
  if (dgamma) fold_param_grads(part, B, C, dgamma, dbeta, accumulate, red, blockIdx.x, gridDim.x);
}

// channel-split form: a workgroup = PX pixels x SPLIT channel groups (PX * SPLIT = 256).  One thread per pixel leaves
// the small maps with 16..64 workgroups, each thread walking 2 C dependent strided loads (47 us for 8.4 MB at 8x8);
// here a thread owns C / SPLIT channels of its pixel (x and gamma*dy stay in registers: one read of each), the two
// row sums meet in LDS.  Consecutive threads are consecutive pixels, so loads stay coalesced.
template <int C, int SPLIT>
__global__ __launch_bounds__(256) void ln_c_bwd_dx_split(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ stats, int HW, long pixels,
                                                         const float* __restrict__ gamma, float* __restrict__ dx,
                                                         const float* __restrict__ part, int B, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, int accumulate, const float* __restrict__ addp) {
  constexpr int PX = 256 / SPLIT, CPT = C / SPLIT;
  static_assert(C % SPLIT == 0 && 256 % SPLIT == 0, "bad split");
  __shared__ float red[16];
  __shared__ float ps[2][SPLIT][PX];
  const int pi = threadIdx.x % PX, g = threadIdx.x / PX;
  const long p = (long)blockIdx.x * PX + pi;
  const bool live = p < pixels;
  float gv[CPT], xh[CPT];
  float s1 = 0.f, s2 = 0.f, rstd = 0.f;
  long off = 0;
  if (live) {
    const long b = p / HW; const int l = p % HW;
    off = b * (long)C * HW + l;
    const float mean = stats[2 * p];
    rstd = stats[2 * p + 1];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = g + i * SPLIT;
      gv[i] = gamma[c] * dy[off + (long)c * HW];
      xh[i] = (x[off + (long)c * HW] - mean) * rstd;
      s1 += gv[i]; s2 += gv[i] * xh[i];
    }
  }
  ps[0][g][pi] = s1; ps[1][g][pi] = s2;
  __syncthreads();
  if (live) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int k = 0; k < SPLIT; ++k) { t1 += ps[0][k][pi]; t2 += ps[1][k][pi]; }
    const float m1 = t1 / (float)C, m2 = t2 / (float)C;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const long o = off + (long)(g + i * SPLIT) * HW;
      dx[o] = rstd * (gv[i] - m1 - xh[i] * m2) + (addp ? addp[o] : 0.f);
    }
  }
  if (dgamma) fold_param_grads(part, B, C, dgamma, dbeta, accumulate, red, blockIdx.x, gridDim.x);
}

// one wave per (b, c) plane: part[b][0][c] = sum_l dy*xhat, part[b][1][c] = sum_l dy
__global__ __launch_bounds__(256) void ln_c_bwd_plane(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ stats, int C, int HW, long planes,
                                                      float* __restrict__ part) {
  const long pl = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (pl >= planes) return;
  const long b = pl / C;
  const long base = pl * HW;
  float s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < HW; i += 64) {
    const long p = b * HW + i;
    const float xh = (x[base + i] - stats[2 * p]) * stats[2 * p + 1];
    const float d = dy[base + i];
    s1 += d * xh; s2 += d;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { const int c = pl % C; part[(2 * b) * C + c] = s1; part[(2 * b + 1) * C + c] = s2; }
}

static inline int gs_grid(long total, int block = 256) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace afd
using namespace afd;

static int g_gn_bwd_mode = 0;      // afd_debug_norm_path 0 / 1: the sample-resident GroupNorm backward by rule / never (plane pass + apply pass)

extern "C" {

int afd_colsum2(const float* in, float* out_a, float* out_b, int rows, int C, int accumulate, afd_stream_t st);
int afd_debug_norm_path(int mode) { g_gn_bwd_mode = mode == 1 ? 1 : 0; return AFD_OK; }

int afd_groupnorm1_fwd(const float* x, float* y, float* stats_out, int B, int C, int HW, float eps,
                       const float* gamma, const float* beta, const float* res, int act, const float* emb,
                       afd_stream_t st) {
  AFD_REQUIRE(x && stats_out && B > 0 && C > 0 && HW > 0, "afd_groupnorm1_fwd: bad argument");
  AFD_REQUIRE(!y || (gamma && beta), "afd_groupnorm1_fwd: gamma/beta required when y is written");
  AFD_REQUIRE(act == 0 || act == 1, "afd_groupnorm1_fwd: act must be 0 or 1");
  hipStream_t s = as_stream(st);
  const long n = (long)C * HW;
  const bool vec = (HW % 4 == 0) && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0);
#define AFD_GN_LAUNCH(NCH, T) hipLaunchKernelGGL(gn_fwd_reg<NCH>, dim3(B), dim3(T), 0, s, x, y, stats_out, C, HW, eps, gamma, beta, res, act, emb)
  if (vec && n <= 65536) {
    const long n4 = n / 4;
    if (n4 <= 256) AFD_GN_LAUNCH(1, 256);
    else if (n4 <= 512) AFD_GN_LAUNCH(2, 256);
    else if (n4 <= 1024) AFD_GN_LAUNCH(4, 256);
    else if (n4 <= 2048) AFD_GN_LAUNCH(8, 256);
    else if (n4 <= 4096) AFD_GN_LAUNCH(4, 1024);
    else if (n4 <= 8192) AFD_GN_LAUNCH(8, 1024);
    else AFD_GN_LAUNCH(16, 1024);
  } else {
    hipLaunchKernelGGL(gn_fwd_loop, dim3(B), dim3(1024), 0, s, x, y, stats_out, C, HW, eps, gamma, beta, res, act, emb);
  }
#undef AFD_GN_LAUNCH
  return check_launch("afd_groupnorm1_fwd");
}

int afd_groupnorm1_bwd(const float* x, const float* dy, const float* stats, int B, int C, int HW,
                       const float* gamma, const float* beta, const float* res, int act,
                       float* dx, float* dres, float* part, float* demb, int have_partials,
                       float* dgamma, float* dbeta, int accumulate, afd_stream_t st) {
  AFD_REQUIRE(x && dy && stats && gamma && beta && dx && part && B > 0 && C > 0 && HW > 0, "afd_groupnorm1_bwd: bad argument");
  AFD_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "afd_groupnorm1_bwd: dgamma and dbeta come together");
  AFD_REQUIRE(act == 0 || act == 1, "afd_groupnorm1_bwd: act must be 0 or 1");
  AFD_REQUIRE(!have_partials || (!dres && !demb && act == 0 && !res), "afd_groupnorm1_bwd: have_partials only for the plain form");
  hipStream_t s = as_stream(st);
  const long planes = (long)B * C;
  const long n = (long)C * HW;
  AFD_REQUIRE(B <= 65535, "afd_groupnorm1_bwd: batch too large for the grid");
  // sample-resident form: x and dy read once, one launch (+ the column sums when the caller wants dgamma / dbeta here)
  const int G = HW / 4;
  const bool vec = HW % 4 == 0 && (G & (G - 1)) == 0 && G <= 1024 && n <= 32768 && g_gn_bwd_mode == 0 &&
                   ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)res | (uintptr_t)dres) & 15) == 0);
  if (!have_partials && vec) {
    const long n4 = n / 4;
#define AFD_GNB_LAUNCH(NCH, T) hipLaunchKernelGGL(gn_bwd_reg<NCH>, dim3(B), dim3(T), 0, s, x, dy, stats, C, HW, gamma, beta, res, act, dx, dres, part, demb)
    if (n4 <= 256 && G <= 256) AFD_GNB_LAUNCH(1, 256);
    else if (n4 <= 512 && G <= 256) AFD_GNB_LAUNCH(2, 256);
    else if (n4 <= 1024 && G <= 256) AFD_GNB_LAUNCH(4, 256);
    else if (n4 <= 2048) AFD_GNB_LAUNCH(2, 1024);
    else if (n4 <= 4096) AFD_GNB_LAUNCH(4, 1024);
    else AFD_GNB_LAUNCH(8, 1024);
#undef AFD_GNB_LAUNCH
    if (dgamma) return afd_colsum2(part, dgamma, dbeta, B, C, accumulate, st);
    return check_launch("afd_groupnorm1_bwd");
  }
  if (!have_partials)
    hipLaunchKernelGGL(gn_bwd_plane, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, s, x, dy, stats, C, HW, planes, gamma, beta, res, act, dres, part, demb);
  int slices = (int)((n + 8191) / 8192);               // ~32 elements (8 float4) per thread
  if (slices < 1) slices = 1;
  if (slices > 64) slices = 64;
  hipLaunchKernelGGL(gn_bwd_apply, dim3(slices, B), dim3(256), 0, s, x, dy, stats, part, C, HW, gamma, beta, res, act, dx, dgamma, dbeta, accumulate);
  return check_launch("afd_groupnorm1_bwd");
}

int afd_layernorm_c_fwd(const float* x, float* y, float* stats_out, int B, int C, int HW, float eps,
                        const float* gamma, const float* beta, afd_stream_t st) {
  AFD_REQUIRE(x && y && stats_out && gamma && beta && B > 0 && C > 0 && HW > 0, "afd_layernorm_c_fwd: bad argument");
  const long pixels = (long)B * HW;
  const dim3 grid((unsigned)((pixels + 255) / 256));
  hipStream_t s = as_stream(st);
#define AFD_LN_SPLIT(C_, S_) hipLaunchKernelGGL((ln_c_fwd_split<C_, S_>), dim3((unsigned)((pixels + 256 / S_ - 1) / (256 / S_))), dim3(256), 0, s, \
                                               x, y, stats_out, HW, pixels, eps, gamma, beta)
  // (forward: the thread-per-pixel register kernel stays ahead on the large maps: 10.7 vs 12.3 us at 32x32)
#define AFD_LN_BY_PIXELS(C_) do { if (pixels >= 32768) hipLaunchKernelGGL(ln_c_fwd_reg<C_>, grid, dim3(256), 0, s, x, y, stats_out, HW, pixels, eps, gamma, beta); \
                                  else if (pixels >= 8192) AFD_LN_SPLIT(C_, 8); else AFD_LN_SPLIT(C_, 16); } while (0)
  switch (C) {
    case 32:  AFD_LN_BY_PIXELS(32); break;
    case 64:  AFD_LN_BY_PIXELS(64); break;
    case 128: AFD_LN_BY_PIXELS(128); break;
    default:  hipLaunchKernelGGL(ln_c_fwd, grid, dim3(256), 0, s, x, y, stats_out, C, HW, pixels, eps, gamma, beta);
  }
#undef AFD_LN_BY_PIXELS
#undef AFD_LN_SPLIT
  return check_launch("afd_layernorm_c_fwd");
}

int afd_layernorm_c_bwd(const float* x, const float* dy, const float* stats, int B, int C, int HW,
                        const float* gamma, float* dx, const float* add, float* part, float* dgamma, float* dbeta, int accumulate,
                        afd_stream_t st) {
  AFD_REQUIRE(x && dy && stats && gamma && dx && B > 0 && C > 0 && HW > 0, "afd_layernorm_c_bwd: bad argument");
  AFD_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "afd_layernorm_c_bwd: dgamma and dbeta come together");
  AFD_REQUIRE(part || !dgamma, "afd_layernorm_c_bwd: the parameter gradients need the partial buffer");
  hipStream_t s = as_stream(st);
  const long pixels = (long)B * HW, planes = (long)B * C;
  const dim3 grid((unsigned)((pixels + 255) / 256));
  // plane partials first: the dx kernel's tail folds them into dgamma / dbeta (part == NULL: dx only)
  if (part) hipLaunchKernelGGL(ln_c_bwd_plane, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, s, x, dy, stats, C, HW, planes, part);
#define AFD_LN_SPLIT(C_, S_) hipLaunchKernelGGL((ln_c_bwd_dx_split<C_, S_>), dim3((unsigned)((pixels + 256 / S_ - 1) / (256 / S_))), dim3(256), 0, s, \
                                               x, dy, stats, HW, pixels, gamma, dx, part, B, dgamma, dbeta, accumulate, add)
#define AFD_LN_BY_PIXELS(C_) do { if (pixels >= 32768) AFD_LN_SPLIT(C_, 4); else if (pixels >= 8192) AFD_LN_SPLIT(C_, 8); else AFD_LN_SPLIT(C_, 16); } while (0)
  switch (C) {                       // >= 256 workgroups on every map of the UNet (4096 pixels at 4x4: 16 channel groups)
    case 32:  AFD_LN_BY_PIXELS(32); break;
    case 64:  AFD_LN_BY_PIXELS(64); break;
    case 128: AFD_LN_BY_PIXELS(128); break;
    default:  hipLaunchKernelGGL(ln_c_bwd_dx, grid, dim3(256), 0, s, x, dy, stats, C, HW, pixels, gamma, dx, part, B, dgamma, dbeta, accumulate, add);
  }
#undef AFD_LN_BY_PIXELS
#undef AFD_LN_SPLIT
  return check_launch("afd_layernorm_c_bwd");
}

int afd_colsum2(const float* in, float* out_a, float* out_b, int rows, int C, int accumulate, afd_stream_t st);

int afd_layernorm_c_bwd_partials(const float* x, const float* dy, const float* stats, int B, int C, int HW, float* part, afd_stream_t st) {
  AFD_REQUIRE(x && dy && stats && part && B > 0 && C > 0 && HW > 0, "afd_layernorm_c_bwd_partials: bad argument");
  const long planes = (long)B * C;
  hipLaunchKernelGGL(ln_c_bwd_plane, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, as_stream(st), x, dy, stats, C, HW, planes, part);
  return check_launch("afd_layernorm_c_bwd_partials");
}

int afd_layernorm_c_bwd_params(const float* x, const float* dy, const float* stats, int B, int C, int HW,
                               float* part, float* dgamma, float* dbeta, int accumulate, afd_stream_t st) {
  AFD_REQUIRE(x && dy && stats && part && dgamma && dbeta && B > 0 && C > 0 && HW > 0, "afd_layernorm_c_bwd_params: bad argument");
  const long planes = (long)B * C;
  hipLaunchKernelGGL(ln_c_bwd_plane, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, as_stream(st), x, dy, stats, C, HW, planes, part);
  return afd_colsum2(part, dgamma, dbeta, B, C, accumulate, st);
}

}  // extern "C"

// tok.hip -- the token-wise chains of SelfAttention (ddpm_utils.py:68-74) fused around the attention core.
//
//   head  (forward):  h = LN1(x);  qkv = W_in h + b_in                                            (ddpm_utils.py:70-71)
//   tail  (forward):  a = W_o att + b_o + x;  f = LN2(a);  u = W_1 f + b_1;  g = GELU(u);  out = W_2 g + b_2 + a     (:71-73)
//   tail  (backward): du = (W_2^T d_out) * GELU'(u);  df = W_1^T du;  d_a = LN2'(df; a) + d_out;  d_att = W_o^T d_a
//   head  (backward): dh = W_in^T dqkv;  dx = LN1'(dh; x) + d_a
//
// Tokens are pixels of the NCHW tensor, so every Linear is a 1x1 convolution and every chain above is pixel-local: a
// WAVE owns 32 consecutive pixels and carries them through the whole chain in registers.  The tile layout ("T layout":
// TT<NB>, C = 32 NB channels x 32 pixels) is the accumulator layout of v_mfma_f32_32x32x2_f32 -- lane = (pixel l31,
// half h), register r of block j = channel 32 j + (r & 3) + 8 (r >> 2) + 4 h -- and it is ALSO a legal B operand of the
// next product: in k-step (j, r) lane half h contributes channel k_h = 32 j + (r & 3) + 8 (r >> 2) + 4 h, and the weight
// fragment is simply read in that k order (lane (n = l31, h) reads W[n][k_h]; four consecutive r are four consecutive k:
// one ds_read_b128 per four MFMAs, rows padded to C + 4 floats: conflict-free).  So GEMM -> LayerNorm -> GEMM -> GELU ->
// GEMM needs no data movement between lanes; LayerNorm's channel sums are a register sum plus ONE exchange with the
// partner lane (l31, 1 - h).  Activations go global <-> registers directly in T layout (per register two 128-byte runs
// per wave).  Replaces 8 forward and 9 backward launches per attention block (LayerNorm, 1x1 conv, GELU, residual
// kernels) by 2 + 2; the 1x1 weight gradients and LayerNorm parameter gradients stay separate launches (side stream).
//
// Weights sit in LDS for the life of the workgroup: [n][C + 4] per C x C matrix.  C = 128 (3 x 67.6 KB > 160 KB) runs the
// three products of a chain as phases over TWO slots: the third matrix replaces the first behind a barrier.
#include "common.h"
#include <algorithm>
#include <cstdint>
#include <mutex>
#include <set>

namespace afd {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NB> struct TT { f32x16 b[NB]; };

__device__ __forceinline__ constexpr int tch(int j, int r) { return 32 * j + (r & 3) + 8 * (r >> 2); }   // + 4 * half

struct Pix { unsigned b, p, f; bool live; };
__device__ __forceinline__ Pix pix_of(int tile, int l31, unsigned total, unsigned P) {
  Pix q;
  q.f = (unsigned)tile * 32u + (unsigned)l31;
  q.live = q.f < total;
  const unsigned fc = q.live ? q.f : 0u;
  q.b = fc / P;
  q.p = fc - q.b * P;
  return q;
}
// element offset of (pixel, channel 4*half) in a tensor with Cx channels
__device__ __forceinline__ unsigned tbase(const Pix& q, unsigned Cx, unsigned P, int half) {
  return q.b * Cx * P + q.p + 4u * (unsigned)half * P;
}

template <int NB>
__device__ __forceinline__ void t_load(TT<NB>& z, const float* __restrict__ src, unsigned base, unsigned P, bool live) {
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) z.b[j][r] = live ? src[base + (unsigned)tch(j, r) * P] : 0.f;
}
template <int NB>
__device__ __forceinline__ void t_store(const TT<NB>& z, float* __restrict__ dst, unsigned base, unsigned P, bool live) {
  if (!live) return;
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[base + (unsigned)tch(j, r) * P] = z.b[j][r];
}

// one 32-channel output block: acc (+)= W[nb*32 .. +32][:] . z     (Ws: LDS [n][32 KB + 4])
template <int KB>
__device__ __forceinline__ f32x16 t_block(const float* __restrict__ Ws, int nb, int l31, int half, const TT<KB>& z, f32x16 acc) {
  constexpr int WS = 32 * KB + 4;
  const float* __restrict__ wr = Ws + (nb * 32 + l31) * WS + 4 * half;
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(wr + 32 * j + 8 * q);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, z.b[j][4 * q + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, z.b[j][4 * q + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, z.b[j][4 * q + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, z.b[j][4 * q + 3], acc, 0, 0, 0);
    }
  return acc;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
  return a;
}

// weights -> LDS.  forward: Ws[n][k] = w[(n0 + n) * ldw + k0 + k];  transposed (dgrad): Ws[n][k] = w[(k0 + k) * ldw + n0 + n]
// (parameters live at arbitrary 4-byte offsets of the flat buffer: scalar accesses)
template <int K>
__device__ __forceinline__ void stage_w(float* __restrict__ Ws, const float* __restrict__ w, int rows, int ldw, int n0, int k0,
                                        bool transposed) {
  constexpr int WS = K + 4;
  if (!transposed) {
    for (int i = threadIdx.x; i < rows * K; i += 256) {
      const int n = i / K, k = i - n * K;
      Ws[n * WS + k] = w[(long)(n0 + n) * ldw + k0 + k];
    }
  } else {
    // lanes run along k (conflict-free LDS stores; the strided global reads re-use each fetched line for 32 n)
    for (int i = threadIdx.x; i < rows * K; i += 256) {
      const int n = i / K, k = i - n * K;
      Ws[n * WS + k] = w[(long)(k0 + k) * ldw + n0 + n];
    }
  }
}

// LayerNorm over the C channels of each pixel, T layout.  gs / bs: gamma / beta in LDS.
template <int KB>
__device__ __forceinline__ void t_ln_fwd(const TT<KB>& x, const float* __restrict__ gs, const float* __restrict__ bs, int half,
                                         float eps, TT<KB>& y, float& mean, float& rstd) {
  constexpr float invC = 1.0f / (32 * KB);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += x.b[j][r];
  s += __shfl_xor(s, 32, 64);
  mean = s * invC;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float d = x.b[j][r] - mean; q = fmaf(d, d, q); }
  q += __shfl_xor(q, 32, 64);
  rstd = 1.0f / sqrtf(q * invC + eps);
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = tch(j, r) + 4 * half;
      y.b[j][r] = (x.b[j][r] - mean) * rstd * gs[c] + bs[c];
    }
}
// dx = rstd * (g*dy - mean_c(g*dy) - xhat * mean_c(g*dy*xhat)) + add      (x: the LayerNorm input, in place in `dy`)
template <int KB>
__device__ __forceinline__ void t_ln_bwd(TT<KB>& dy, const TT<KB>& x, const float* __restrict__ gs, int half, float mean,
                                         float rstd, const TT<KB>& add) {
  constexpr float invC = 1.0f / (32 * KB);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float gv = gs[tch(j, r) + 4 * half] * dy.b[j][r];
      const float xh = (x.b[j][r] - mean) * rstd;
      dy.b[j][r] = gv;
      s1 += gv;
      s2 = fmaf(gv, xh, s2);
    }
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  const float m1 = s1 * invC, m2 = s2 * invC;
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float xh = (x.b[j][r] - mean) * rstd;
      dy.b[j][r] = rstd * (dy.b[j][r] - m1 - xh * m2) + add.b[j][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// head forward: qkv = W_in LN1(x) + b_in.  grid (pixel-tile groups, N / n_per_wg): the output channels are split over
// workgroups (each recomputes the cheap LayerNorm); blockIdx.y == 0 also writes h = LN1(x) and the statistics.
// ---------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void tok_head_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ h_out,
                                                    float* __restrict__ stats_out, float* __restrict__ y, int B, int N, int P,
                                                    float eps, int n_per_wg) {
  constexpr int KB = C / 32, WS = C + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                       // [n_per_wg][WS]
  float* bsm = Ws + n_per_wg * WS;        // [n_per_wg]
  float* gs = bsm + n_per_wg;             // [C]
  float* bes = gs + C;                    // [C]
  const int nbase = blockIdx.y * n_per_wg;
  stage_w<C>(Ws, w, n_per_wg, C, nbase, 0, false);
  for (int i = threadIdx.x; i < n_per_wg; i += 256) bsm[i] = bias ? bias[nbase + i] : 0.f;
  for (int i = threadIdx.x; i < C; i += 256) { gs[i] = gamma[i]; bes[i] = beta[i]; }
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const unsigned total = (unsigned)B * (unsigned)P;
  const int ntile = (int)((total + 31u) / 32u);
  const int tstride = gridDim.x * 4;
  const bool lead = blockIdx.y == 0;
  const int nblk = n_per_wg / 32;

  TT<KB> xc, xn;
  Pix pc, pn;
  int t = blockIdx.x * 4 + wv;
  if (t < ntile) { pc = pix_of(t, l31, total, P); t_load(xc, x, tbase(pc, C, P, half), P, pc.live); }
  for (; t < ntile; t += tstride) {
    const int tn = t + tstride;
    if (tn < ntile) { pn = pix_of(tn, l31, total, P); t_load(xn, x, tbase(pn, C, P, half), P, pn.live); }
    TT<KB> h;
    float mean, rstd;
    t_ln_fwd<KB>(xc, gs, bes, half, eps, h, mean, rstd);
    if (lead) {
      if (h_out) t_store(h, h_out, tbase(pc, C, P, half), P, pc.live);
      if (stats_out && pc.live && half == 0) { stats_out[2 * pc.f] = mean; stats_out[2 * pc.f + 1] = rstd; }
    }
    const unsigned yb = tbase(pc, N, P, half);
    for (int nb = 0; nb < nblk; ++nb) {
      const f32x16 acc = t_block<KB>(Ws, nb, l31, half, h, zero16());
      if (pc.live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int nl = nb * 32 + tch(0, r) + 4 * half;
          y[yb + (unsigned)(nbase + nb * 32 + tch(0, r)) * (unsigned)P] = acc[r] + bsm[nl];
        }
      }
    }
    xc = xn; pc = pn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// three-matrix chains (tail forward, tail backward, head backward): common skeleton.
// LDS: slots of [C][C+4]; C <= 64: three slots, filled once; C = 128: two slots, the third matrix replaces the first
// behind a barrier in every pass (RELOAD).  All waves of a workgroup run the same number of passes (barriers).
// ---------------------------------------------------------------------------------------------------------------------
template <int C> struct Chain {
  static constexpr int KB = C / 32, WS = C + 4, SLOT = C * WS;
  static constexpr bool RELOAD = C > 64;
  static constexpr int NSLOT = RELOAD ? 2 : 3;
};

// tail forward.  TRAIN: also writes a, LN2 statistics, f, u (pre-GELU) and g (post-GELU) for the backward pass.
template <int C, bool TRAIN>
__global__ __launch_bounds__(256) void tok_tail_fwd(const float* __restrict__ att, const float* __restrict__ xres,
                                                    const float* __restrict__ wo, const float* __restrict__ bo,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ w1, const float* __restrict__ b1,
                                                    const float* __restrict__ w2, const float* __restrict__ b2,
                                                    float* __restrict__ a_out, float* __restrict__ stats_out,
                                                    float* __restrict__ f_out, float* __restrict__ u_out, float* __restrict__ g_out,
                                                    float* __restrict__ out, int B, int P, float eps) {
  using CH = Chain<C>;
  constexpr int KB = CH::KB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s0 = smem;
  float* s1 = smem + CH::SLOT;
  float* s2 = CH::RELOAD ? s0 : smem + 2 * CH::SLOT;
  float* vec = smem + CH::NSLOT * CH::SLOT;           // bo, gamma, beta, b1, b2
  stage_w<C>(s0, wo, C, C, 0, 0, false);
  stage_w<C>(s1, w1, C, C, 0, 0, false);
  if (!CH::RELOAD) stage_w<C>(s2, w2, C, C, 0, 0, false);
  for (int i = threadIdx.x; i < C; i += 256) {
    vec[i] = bo[i]; vec[C + i] = gamma[i]; vec[2 * C + i] = beta[i]; vec[3 * C + i] = b1[i]; vec[4 * C + i] = b2[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const unsigned total = (unsigned)B * (unsigned)P;
  const int ntile = (int)((total + 31u) / 32u);
  const int tstride = gridDim.x * 4;
  const int t0 = blockIdx.x * 4;
  const int npass = t0 < ntile ? (ntile - t0 + tstride - 1) / tstride : 0;
  for (int pass = 0; pass < npass; ++pass) {
    const int t = t0 + wv + pass * tstride;
    const bool active = t < ntile;                    // wave-uniform
    if (CH::RELOAD && pass > 0) { __syncthreads(); stage_w<C>(s0, wo, C, C, 0, 0, false); __syncthreads(); }
    Pix pc = pix_of(active ? t : 0, l31, total, P);
    pc.live = pc.live && active;
    const unsigned base = tbase(pc, C, P, half);
    TT<KB> a, z;
    if (active) {
      TT<KB> xr;
      t_load(z, att, base, P, pc.live);
      t_load(xr, xres, base, P, pc.live);
#pragma unroll
      for (int nb = 0; nb < KB; ++nb) {
        const f32x16 acc = t_block<KB>(s0, nb, l31, half, z, zero16());
#pragma unroll
        for (int r = 0; r < 16; ++r) a.b[nb][r] = acc[r] + vec[nb * 32 + tch(0, r) + 4 * half] + xr.b[nb][r];
      }
      float mean, rstd;
      t_ln_fwd<KB>(a, vec + C, vec + 2 * C, half, eps, z, mean, rstd);          // z = f
      if (TRAIN) {
        t_store(a, a_out, base, P, pc.live);
        t_store(z, f_out, base, P, pc.live);
        if (pc.live && half == 0) { stats_out[2 * pc.f] = mean; stats_out[2 * pc.f + 1] = rstd; }
      }
    }
    if (CH::RELOAD) { __syncthreads(); stage_w<C>(s0, w2, C, C, 0, 0, false); }
    TT<KB> g;
    if (active) {
#pragma unroll
      for (int nb = 0; nb < KB; ++nb) {
        const f32x16 acc = t_block<KB>(s1, nb, l31, half, z, zero16());
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float u = acc[r] + vec[3 * C + nb * 32 + tch(0, r) + 4 * half];
          if (TRAIN && pc.live) u_out[base + (unsigned)tch(nb, r) * (unsigned)P] = u;
          g.b[nb][r] = gelu_erf(u);
        }
      }
      if (TRAIN) t_store(g, g_out, base, P, pc.live);
    }
    if (CH::RELOAD) __syncthreads();
    if (active) {
#pragma unroll
      for (int nb = 0; nb < KB; ++nb) {
        const f32x16 acc = t_block<KB>(s2, nb, l31, half, g, zero16());
        if (pc.live) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            out[base + (unsigned)tch(nb, r) * (unsigned)P] = acc[r] + vec[4 * C + nb * 32 + tch(0, r) + 4 * half] + a.b[nb][r];
        }
      }
    }
  }
}

// tail backward: du = (W2^T d_out) * GELU'(u);  df = W1^T du;  d_a = LN2'(df; a) + d_out;  d_att = Wo^T d_a
template <int C>
__global__ __launch_bounds__(256) void tok_tail_bwd(const float* __restrict__ dout, const float* __restrict__ u_in,
                                                    const float* __restrict__ a_in, const float* __restrict__ stats,
                                                    const float* __restrict__ gamma, const float* __restrict__ w2,
                                                    const float* __restrict__ w1, const float* __restrict__ wo,
                                                    float* __restrict__ du_out, float* __restrict__ df_out,
                                                    float* __restrict__ da_out, float* __restrict__ datt_out, int B, int P) {
  using CH = Chain<C>;
  constexpr int KB = CH::KB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s0 = smem;
  float* s1 = smem + CH::SLOT;
  float* s2 = CH::RELOAD ? s0 : smem + 2 * CH::SLOT;
  float* gs = smem + CH::NSLOT * CH::SLOT;
  stage_w<C>(s0, w2, C, C, 0, 0, true);
  stage_w<C>(s1, w1, C, C, 0, 0, true);
  if (!CH::RELOAD) stage_w<C>(s2, wo, C, C, 0, 0, true);
  for (int i = threadIdx.x; i < C; i += 256) gs[i] = gamma[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const unsigned total = (unsigned)B * (unsigned)P;
  const int ntile = (int)((total + 31u) / 32u);
  const int tstride = gridDim.x * 4;
  const int t0 = blockIdx.x * 4;
  const int npass = t0 < ntile ? (ntile - t0 + tstride - 1) / tstride : 0;
  for (int pass = 0; pass < npass; ++pass) {
    const int t = t0 + wv + pass * tstride;
    const bool active = t < ntile;
    if (CH::RELOAD && pass > 0) { __syncthreads(); stage_w<C>(s0, w2, C, C, 0, 0, true); __syncthreads(); }
    Pix pc = pix_of(active ? t : 0, l31, total, P);
    pc.live = pc.live && active;
    const unsigned base = tbase(pc, C, P, half);
    TT<KB> dy, z;
    if (active) {
      TT<KB> uu;
      t_load(dy, dout, base, P, pc.live);
      t_load(uu, u_in, base, P, pc.live);
#pragma unroll
      for (int nb = 0; nb < KB; ++nb) {
        const f32x16 acc = t_block<KB>(s0, nb, l31, half, dy, zero16());
#pragma unroll
        for (int r = 0; r < 16; ++r) z.b[nb][r] = acc[r] * gelu_erf_grad(uu.b[nb][r]);
      }
      t_store(z, du_out, base, P, pc.live);                                       // du
    }
    if (CH::RELOAD) { __syncthreads(); stage_w<C>(s0, wo, C, C, 0, 0, true); }
    if (active) {
      TT<KB> df, aa;
      t_load(aa, a_in, base, P, pc.live);
#pragma unroll
      for (int nb = 0; nb < KB; ++nb) df.b[nb] = t_block<KB>(s1, nb, l31, half, z, zero16());
      t_store(df, df_out, base, P, pc.live);
      const float mean = pc.live ? stats[2 * pc.f] : 0.f, rstd = pc.live ? stats[2 * pc.f + 1] : 0.f;
      t_ln_bwd<KB>(df, aa, gs, half, mean, rstd, dy);                             // df := d_a
      t_store(df, da_out, base, P, pc.live);
      z = df;
    }
    if (CH::RELOAD) __syncthreads();
    if (active) {
#pragma unroll
      for (int nb = 0; nb < KB; ++nb) {
        const f32x16 acc = t_block<KB>(s2, nb, l31, half, z, zero16());
        if (pc.live) {
#pragma unroll
          for (int r = 0; r < 16; ++r) datt_out[base + (unsigned)tch(nb, r) * (unsigned)P] = acc[r];
        }
      }
    }
  }
}

// head backward: dh = W_in^T dqkv (the three C x C blocks of W_in as the three matrices);  dx = LN1'(dh; x) + d_res
template <int C>
__global__ __launch_bounds__(256) void tok_head_bwd(const float* __restrict__ dqkv, const float* __restrict__ x,
                                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                                    const float* __restrict__ w_in, const float* __restrict__ dres,
                                                    float* __restrict__ dh_out, float* __restrict__ dx_out, int B, int P) {
  using CH = Chain<C>;
  constexpr int KB = CH::KB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s0 = smem;
  float* s1 = smem + CH::SLOT;
  float* s2 = CH::RELOAD ? s0 : smem + 2 * CH::SLOT;
  float* gs = smem + CH::NSLOT * CH::SLOT;
  // block s of W_in is rows [sC, (s+1)C): Ws[n][k] = w_in[(sC + k) * C + n]
  stage_w<C>(s0, w_in, C, C, 0, 0, true);
  stage_w<C>(s1, w_in, C, C, 0, C, true);
  if (!CH::RELOAD) stage_w<C>(s2, w_in, C, C, 0, 2 * C, true);
  for (int i = threadIdx.x; i < C; i += 256) gs[i] = gamma[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const unsigned total = (unsigned)B * (unsigned)P;
  const int ntile = (int)((total + 31u) / 32u);
  const int tstride = gridDim.x * 4;
  const int t0 = blockIdx.x * 4;
  const int npass = t0 < ntile ? (ntile - t0 + tstride - 1) / tstride : 0;
  for (int pass = 0; pass < npass; ++pass) {
    const int t = t0 + wv + pass * tstride;
    const bool active = t < ntile;
    if (CH::RELOAD && pass > 0) { __syncthreads(); stage_w<C>(s0, w_in, C, C, 0, 0, true); __syncthreads(); }
    Pix pc = pix_of(active ? t : 0, l31, total, P);
    pc.live = pc.live && active;
    const unsigned base = tbase(pc, C, P, half);
    const unsigned qb = tbase(pc, 3 * C, P, half);
    TT<KB> dh, z;
    if (active) {
      t_load(z, dqkv, qb, P, pc.live);
#pragma unroll
      for (int nb = 0; nb < KB; ++nb) dh.b[nb] = t_block<KB>(s0, nb, l31, half, z, zero16());
    }
    if (CH::RELOAD) { __syncthreads(); stage_w<C>(s0, w_in, C, C, 0, 2 * C, true); }
    if (active) {
      t_load(z, dqkv, qb + (unsigned)C * (unsigned)P, P, pc.live);
#pragma unroll
      for (int nb = 0; nb < KB; ++nb) dh.b[nb] = t_block<KB>(s1, nb, l31, half, z, dh.b[nb]);
    }
    if (CH::RELOAD) __syncthreads();
    if (active) {
      t_load(z, dqkv, qb + 2u * (unsigned)C * (unsigned)P, P, pc.live);
#pragma unroll
      for (int nb = 0; nb < KB; ++nb) dh.b[nb] = t_block<KB>(s2, nb, l31, half, z, dh.b[nb]);
      if (dh_out) t_store(dh, dh_out, base, P, pc.live);
      TT<KB> xx, rr;
      t_load(xx, x, base, P, pc.live);
      t_load(rr, dres, base, P, pc.live);
      const float mean = pc.live ? stats[2 * pc.f] : 0.f, rstd = pc.live ? stats[2 * pc.f + 1] : 0.f;
      t_ln_bwd<KB>(dh, xx, gs, half, mean, rstd, rr);
      t_store(dh, dx_out, base, P, pc.live);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// "wide" forms for the small maps (C = 128: 8x8 and 4x4; C = 64 below 1024 tiles).  There a layer has only 128 .. 512
// pixel tiles, a wave carrying a tile alone through C/32 output blocks per product runs 6.8 us of dependent MFMAs per
// product on one SIMD while three quarters of the chip idle, and staging 3 x 64 KB of weights per workgroup is most of
// the launch.  Here the C/32 output blocks of a tile are split over C/32 WAVES (a workgroup = 4 / (C/32) tiles): a wave
// computes ONE 32-channel block per product (64 MFMAs at C = 128), its weight rows come straight from global memory / L2
// into registers in A-operand order (16 KB per wave and product, the next product's rows in flight behind the current
// one; no LDS image, no staging barriers, no two-slot reload), and the waves of a tile exchange their blocks through a
// 16-KB LDS tile [c][pixel] to rebuild the full T-layout operand of the next product / the LayerNorm sums.
// ---------------------------------------------------------------------------------------------------------------------
template <int KB> struct WA { float4 v[4 * KB]; };

// A operand rows of a row-major [N][ldw] matrix: lane (n = l31, h) holds W[row0 + l31][k0 + 32 j + 8 q + 4 h + 0..3]
template <int KB>
__device__ __forceinline__ void wa_load(WA<KB>& a, const float* __restrict__ w, int row0, int ldw, int k0, int l31, int half) {
  const float4* __restrict__ p = reinterpret_cast<const float4*>(w + (long)(row0 + l31) * ldw + k0 + 4 * half);
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) a.v[4 * j + q] = p[8 * j + 2 * q];
}
// transposed (dgrad): A[n][k] = W[k0 + k][n0 + n]; for fixed k the 32 lanes of a half read one 128-byte run
template <int KB>
__device__ __forceinline__ void wa_load_t(WA<KB>& a, const float* __restrict__ w, int n0, int ldw, int k0, int l31, int half) {
  const float* __restrict__ p = w + (long)(k0 + 4 * half) * ldw + n0 + l31;
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* __restrict__ r0 = p + (long)(32 * j + 8 * q) * ldw;
      a.v[4 * j + q] = make_float4(r0[0], r0[ldw], r0[2 * ldw], r0[3 * ldw]);
    }
}
template <int KB>
__device__ __forceinline__ f32x16 wa_block(const WA<KB>& a, const TT<KB>& z, f32x16 acc) {
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[4 * j + q].x, z.b[j][4 * q + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[4 * j + q].y, z.b[j][4 * q + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[4 * j + q].z, z.b[j][4 * q + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[4 * j + q].w, z.b[j][4 * q + 3], acc, 0, 0, 0);
    }
  return acc;
}
// exchange tile X[c][32 pixels]: a wave deposits its 32-channel block, every wave of the tile reads the whole tile
__device__ __forceinline__ void x_put(float* __restrict__ X, int nbk, int l31, int half, const f32x16& blk) {
#pragma unroll
  for (int r = 0; r < 16; ++r) X[(nbk * 32 + tch(0, r) + 4 * half) * 32 + l31] = blk[r];
}
template <int KB>
__device__ __forceinline__ void x_get(const float* __restrict__ X, int l31, int half, TT<KB>& z) {
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) z.b[j][r] = X[(tch(j, r) + 4 * half) * 32 + l31];
}
__device__ __forceinline__ f32x16 b_load(const float* __restrict__ src, unsigned base, unsigned P, bool live) {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = live ? src[base + (unsigned)tch(0, r) * P] : 0.f;
  return v;
}
__device__ __forceinline__ void b_store(const f32x16& v, float* __restrict__ dst, unsigned base, unsigned P, bool live) {
  if (!live) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[base + (unsigned)tch(0, r) * P] = v[r];
}
// channel sums of LayerNorm over a full tile
template <int KB>
__device__ __forceinline__ void t_ln_stats(const TT<KB>& x, float eps, float& mean, float& rstd) {
  constexpr float invC = 1.0f / (32 * KB);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += x.b[j][r];
  s += __shfl_xor(s, 32, 64);
  mean = s * invC;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float d = x.b[j][r] - mean; q = fmaf(d, d, q); }
  q += __shfl_xor(q, 32, 64);
  rstd = 1.0f / sqrtf(q * invC + eps);
}
// LayerNorm backward on a full tile: dy := rstd * (g dy - m1 - xhat m2); returns m1, m2 for the per-block restatement
template <int KB>
__device__ __forceinline__ void t_ln_bwd_core(TT<KB>& dy, const TT<KB>& x, const float* __restrict__ gs, int half, float mean,
                                              float rstd, float& m1, float& m2) {
  constexpr float invC = 1.0f / (32 * KB);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float gv = gs[tch(j, r) + 4 * half] * dy.b[j][r];
      const float xh = (x.b[j][r] - mean) * rstd;
      dy.b[j][r] = gv;
      s1 += gv;
      s2 = fmaf(gv, xh, s2);
    }
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  m1 = s1 * invC; m2 = s2 * invC;
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float xh = (x.b[j][r] - mean) * rstd;
      dy.b[j][r] = rstd * (dy.b[j][r] - m1 - xh * m2);
    }
}

template <int C> struct Wide {
  static constexpr int KB = C / 32, NS = C / 32, TPW = 4 / NS;      // waves per tile, tiles per workgroup
};

// head forward, wide: one wave per (tile, NBW consecutive 32-channel blocks of qkv); every wave normalises its tile itself
// (NBW = 1: 3C/32 waves per tile, the 4x4 maps; NBW = 3: C/32 waves per tile, one third of the redundant LayerNorms)
template <int C, int NBW>
__global__ __launch_bounds__(256) void tok_head_fwd_wide(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ h_out,
                                                         float* __restrict__ stats_out, float* __restrict__ y, int B, int P,
                                                         float eps) {
  constexpr int KB = C / 32, N = 3 * C, NGRP = N / 32 / NBW;
  __shared__ float vec[5 * C];                       // gamma, beta, bias[3C]
  for (int i = threadIdx.x; i < C; i += 256) { vec[i] = gamma[i]; vec[C + i] = beta[i]; }
  for (int i = threadIdx.x; i < N; i += 256) vec[2 * C + i] = bias ? bias[i] : 0.f;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const unsigned total = (unsigned)B * (unsigned)P;
  const int ntile = (int)((total + 31u) / 32u);
  const long gw = (long)blockIdx.x * 4 + wv;
  const int tile = (int)(gw / NGRP), nb0 = (int)(gw - (long)tile * NGRP) * NBW;
  const bool active = tile < ntile;
  Pix pc = pix_of(active ? tile : 0, l31, total, P);
  pc.live = pc.live && active;
  TT<KB> xt, h;
  WA<KB> wa, wb;
  t_load(xt, x, tbase(pc, C, P, half), P, pc.live);
  wa_load(wa, w, nb0 * 32, C, 0, l31, half);
  __syncthreads();
  float mean, rstd;
  t_ln_fwd<KB>(xt, vec, vec + C, half, eps, h, mean, rstd);
  if (nb0 == 0) {
    if (h_out) t_store(h, h_out, tbase(pc, C, P, half), P, pc.live);
    if (stats_out && pc.live && half == 0) { stats_out[2 * pc.f] = mean; stats_out[2 * pc.f + 1] = rstd; }
  }
  const unsigned yb = tbase(pc, N, P, half);
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int nb = nb0 + i;
    if (i + 1 < NBW) wa_load((i & 1) ? wa : wb, w, (nb + 1) * 32, C, 0, l31, half);     // the next block's rows in flight
    f32x16 acc = wa_block<KB>((i & 1) ? wb : wa, h, zero16());
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += vec[2 * C + nb * 32 + tch(0, r) + 4 * half];
    b_store(acc, y, yb + (unsigned)(nb * 32) * (unsigned)P, P, pc.live);
  }
}

template <int C, bool TRAIN>
__global__ __launch_bounds__(256) void tok_tail_fwd_wide(const float* __restrict__ att, const float* __restrict__ xres,
                                                         const float* __restrict__ wo, const float* __restrict__ bo,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         float* __restrict__ a_out, float* __restrict__ stats_out,
                                                         float* __restrict__ f_out, float* __restrict__ u_out,
                                                         float* __restrict__ g_out, float* __restrict__ out, int B, int P, float eps) {
  using WD = Wide<C>;
  constexpr int KB = WD::KB;
  __shared__ float X[WD::TPW][C * 32];
  __shared__ float vec[5 * C];                       // bo, gamma, beta, b1, b2
  for (int i = threadIdx.x; i < C; i += 256) {
    vec[i] = bo[i]; vec[C + i] = gamma[i]; vec[2 * C + i] = beta[i]; vec[3 * C + i] = b1[i]; vec[4 * C + i] = b2[i];
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const int ts = wv / WD::NS, nbk = wv % WD::NS;
  const unsigned total = (unsigned)B * (unsigned)P;
  const int ntile = (int)((total + 31u) / 32u);
  const int tile = blockIdx.x * WD::TPW + ts;
  const bool active = tile < ntile;
  Pix pc = pix_of(active ? tile : 0, l31, total, P);
  pc.live = pc.live && active;
  const unsigned base = tbase(pc, C, P, half), bb = base + (unsigned)(nbk * 32) * (unsigned)P;
  const int cb = nbk * 32 + 4 * half;                // + tch(0, r): this lane's channels of its block
  float* Xt = X[ts];
  TT<KB> z;
  WA<KB> wa, wb;
  t_load(z, att, base, P, pc.live);
  const f32x16 xr = b_load(xres, bb, P, pc.live);
  wa_load(wa, wo, nbk * 32, C, 0, l31, half);
  __syncthreads();                                   // vec
  f32x16 ab = wa_block<KB>(wa, z, zero16());
  wa_load(wb, w1, nbk * 32, C, 0, l31, half);
#pragma unroll
  for (int r = 0; r < 16; ++r) ab[r] += vec[cb + tch(0, r)] + xr[r];
  x_put(Xt, nbk, l31, half, ab);
  if (TRAIN) b_store(ab, a_out, bb, P, pc.live);
  __syncthreads();
  x_get<KB>(Xt, l31, half, z);                       // a, whole tile
  float mean, rstd;
  t_ln_stats<KB>(z, eps, mean, rstd);
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = tch(j, r) + 4 * half;
      z.b[j][r] = (z.b[j][r] - mean) * rstd * vec[C + c] + vec[2 * C + c];
    }
  if (TRAIN) {
    f32x16 fb;
#pragma unroll
    for (int r = 0; r < 16; ++r) fb[r] = (ab[r] - mean) * rstd * vec[C + cb + tch(0, r)] + vec[2 * C + cb + tch(0, r)];
    b_store(fb, f_out, bb, P, pc.live);
    if (nbk == 0 && pc.live && half == 0) { stats_out[2 * pc.f] = mean; stats_out[2 * pc.f + 1] = rstd; }
  }
  __syncthreads();                                   // every wave has read a: the exchange tile is free
  f32x16 gb = wa_block<KB>(wb, z, zero16());
  wa_load(wa, w2, nbk * 32, C, 0, l31, half);
#pragma unroll
  for (int r = 0; r < 16; ++r) gb[r] += vec[3 * C + cb + tch(0, r)];
  if (TRAIN) b_store(gb, u_out, bb, P, pc.live);
#pragma unroll
  for (int r = 0; r < 16; ++r) gb[r] = gelu_erf(gb[r]);
  if (TRAIN) b_store(gb, g_out, bb, P, pc.live);
  x_put(Xt, nbk, l31, half, gb);
  __syncthreads();
  x_get<KB>(Xt, l31, half, z);                       // g, whole tile
  f32x16 ob = wa_block<KB>(wa, z, zero16());
#pragma unroll
  for (int r = 0; r < 16; ++r) ob[r] += vec[4 * C + cb + tch(0, r)] + ab[r];
  b_store(ob, out, bb, P, pc.live);
}

template <int C>
__global__ __launch_bounds__(256) void tok_tail_bwd_wide(const float* __restrict__ dout, const float* __restrict__ u_in,
                                                         const float* __restrict__ a_in, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ w2,
                                                         const float* __restrict__ w1, const float* __restrict__ wo,
                                                         float* __restrict__ du_out, float* __restrict__ df_out,
                                                         float* __restrict__ da_out, float* __restrict__ datt_out, int B, int P) {
  using WD = Wide<C>;
  constexpr int KB = WD::KB;
  __shared__ float X[WD::TPW][C * 32];
  __shared__ float gs[C];
  for (int i = threadIdx.x; i < C; i += 256) gs[i] = gamma[i];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const int ts = wv / WD::NS, nbk = wv % WD::NS;
  const unsigned total = (unsigned)B * (unsigned)P;
  const int ntile = (int)((total + 31u) / 32u);
  const int tile = blockIdx.x * WD::TPW + ts;
  const bool active = tile < ntile;
  Pix pc = pix_of(active ? tile : 0, l31, total, P);
  pc.live = pc.live && active;
  const unsigned base = tbase(pc, C, P, half), bb = base + (unsigned)(nbk * 32) * (unsigned)P;
  const int cb = nbk * 32 + 4 * half;
  float* Xt = X[ts];
  TT<KB> z, aa;
  WA<KB> wa, wb;
  t_load(z, dout, base, P, pc.live);
  const f32x16 ub = b_load(u_in, bb, P, pc.live);
  wa_load_t(wa, w2, nbk * 32, C, 0, l31, half);
  f32x16 dub = wa_block<KB>(wa, z, zero16());
  wa_load_t(wb, w1, nbk * 32, C, 0, l31, half);
#pragma unroll
  for (int r = 0; r < 16; ++r) dub[r] *= gelu_erf_grad(ub[r]);
  b_store(dub, du_out, bb, P, pc.live);
  x_put(Xt, nbk, l31, half, dub);
  __syncthreads();                                   // (also: gs)
  x_get<KB>(Xt, l31, half, z);                       // du, whole tile
  const f32x16 dfb = wa_block<KB>(wb, z, zero16());
  wa_load_t(wa, wo, nbk * 32, C, 0, l31, half);
  b_store(dfb, df_out, bb, P, pc.live);
  t_load(aa, a_in, base, P, pc.live);
  __syncthreads();                                   // every wave has read du
  x_put(Xt, nbk, l31, half, dfb);
  __syncthreads();
  x_get<KB>(Xt, l31, half, z);                       // df, whole tile
  const float mean = pc.live ? stats[2 * pc.f] : 0.f, rstd = pc.live ? stats[2 * pc.f + 1] : 0.f;
  float m1, m2;
  t_ln_bwd_core<KB>(z, aa, gs, half, mean, rstd, m1, m2);
  t_load(aa, dout, base, P, pc.live);                // the residual branch: d_a = LN'(df) + d_out
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) z.b[j][r] += aa.b[j][r];
  {                                                  // this wave's block of d_a, from its own block registers
    const f32x16 ab = b_load(a_in, bb, P, pc.live), dob = b_load(dout, bb, P, pc.live);
    f32x16 dab;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float xh = (ab[r] - mean) * rstd;
      dab[r] = rstd * (gs[cb + tch(0, r)] * dfb[r] - m1 - xh * m2) + dob[r];
    }
    b_store(dab, da_out, bb, P, pc.live);
  }
  const f32x16 db = wa_block<KB>(wa, z, zero16());
  b_store(db, datt_out, bb, P, pc.live);
}

template <int C>
__global__ __launch_bounds__(256) void tok_head_bwd_wide(const float* __restrict__ dqkv, const float* __restrict__ x,
                                                         const float* __restrict__ stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ w_in, const float* __restrict__ dres,
                                                         float* __restrict__ dh_out, float* __restrict__ dx_out, int B, int P) {
  using WD = Wide<C>;
  constexpr int KB = WD::KB;
  __shared__ float X[WD::TPW][C * 32];
  __shared__ float gs[C];
  for (int i = threadIdx.x; i < C; i += 256) gs[i] = gamma[i];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const int ts = wv / WD::NS, nbk = wv % WD::NS;
  const unsigned total = (unsigned)B * (unsigned)P;
  const int ntile = (int)((total + 31u) / 32u);
  const int tile = blockIdx.x * WD::TPW + ts;
  const bool active = tile < ntile;
  Pix pc = pix_of(active ? tile : 0, l31, total, P);
  pc.live = pc.live && active;
  const unsigned base = tbase(pc, C, P, half), bb = base + (unsigned)(nbk * 32) * (unsigned)P;
  const unsigned qb = tbase(pc, 3 * C, P, half);
  const int cb = nbk * 32 + 4 * half;
  float* Xt = X[ts];
  TT<KB> z;
  WA<KB> wa;
  f32x16 dhb = zero16();
#pragma unroll
  for (int sct = 0; sct < 3; ++sct) {                // the q, k and v row blocks of W_in
    t_load(z, dqkv, qb + (unsigned)(sct * C) * (unsigned)P, P, pc.live);
    wa_load_t(wa, w_in, nbk * 32, C, sct * C, l31, half);
    dhb = wa_block<KB>(wa, z, dhb);
  }
  if (dh_out) b_store(dhb, dh_out, bb, P, pc.live);
  x_put(Xt, nbk, l31, half, dhb);
  TT<KB> xx;
  t_load(xx, x, base, P, pc.live);
  __syncthreads();                                   // (also: gs)
  x_get<KB>(Xt, l31, half, z);                       // dh, whole tile
  const float mean = pc.live ? stats[2 * pc.f] : 0.f, rstd = pc.live ? stats[2 * pc.f + 1] : 0.f;
  float m1, m2;
  t_ln_bwd_core<KB>(z, xx, gs, half, mean, rstd, m1, m2);
  const f32x16 xb = b_load(x, bb, P, pc.live), rb = b_load(dres, bb, P, pc.live);
  f32x16 dxb;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float xh = (xb[r] - mean) * rstd;
    dxb[r] = rstd * (gs[cb + tch(0, r)] * dhb[r] - m1 - xh * m2) + rb[r];
  }
  b_store(dxb, dx_out, bb, P, pc.live);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static bool tok_ok(int C) { return C == 32 || C == 64 || C == 128; }
static size_t chain_lds(int C, int nvec) {
  const int nslot = C > 64 ? 2 : 3;
  return sizeof(float) * ((size_t)nslot * C * (C + 4) + (size_t)nvec * C);
}
// kernels that need more than 64 KB of dynamic LDS opt in once per (device, kernel): common.h lds_opt_in
template <class K> static int set_lds(K kern, size_t lds) { return lds_opt_in(kern, lds); }
static int g_tok_max_wg = 0;       // test hook (afd_debug_tok_grid): cap on the workgroups of a launch, 0 = by the rule
static int g_tok_path = 0;         // test hook (afd_debug_tok_path): 0 = by the rule, 1 = never the wide forms, 2 = wide wherever they exist
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// wide forms: C = 128 always, C = 64 for layers of fewer than 1024 pixel tiles (they exist for C in {64, 128})
static bool use_wide(int C, long ntile) {
  if (C != 64 && C != 128) return false;
  if (g_tok_max_wg > 0 || g_tok_path == 1) return false;
  if (g_tok_path == 2) return true;
  return C == 128 || ntile < 1024;
}
static unsigned chain_grid(int C, long ntile) {
  const long wg = (ntile + 3) / 4;
  const long cap = g_tok_max_wg > 0 ? g_tok_max_wg : (C > 64 ? 8192 : 1024);
  return (unsigned)std::min<long>(wg, cap);
}

}  // namespace afd

using namespace afd;

extern "C" {

int afd_tok_supported(int C) { return tok_ok(C) ? 1 : 0; }
int afd_debug_tok_grid(int max_workgroups) { g_tok_max_wg = max_workgroups; return AFD_OK; }
int afd_debug_tok_path(int mode) { g_tok_path = mode; return AFD_OK; }

int afd_tok_head_fwd(const float* x, const float* gamma, const float* beta, const float* w, const float* bias, float* h_out,
                     float* stats_out, float* qkv, int B, int C, int P, float eps, afd_stream_t st) {
  AFD_REQUIRE(x && gamma && beta && w && qkv && B > 0 && P > 0, "afd_tok_head_fwd: NULL pointer or empty shape");
  AFD_REQUIRE(tok_ok(C), "afd_tok_head_fwd: C=%d not covered (32, 64, 128)", C);
  AFD_REQUIRE((long)B * P * 3 * C < (1L << 31), "afd_tok_head_fwd: tensor too large for 32-bit offsets");
  const int N = 3 * C;
  const long ntile = ((long)B * P + 31) / 32, wg_tiles = (ntile + 3) / 4;
  // wide head: one wave per (tile, 32-channel block), 3C/32 redundant LayerNorms per tile -- pays only while the layer has too
  // few tiles to fill the chip otherwise (measured at B = 256: 128 -> 384 @ 4x4 15.8 -> 13.8 us, @ 8x8 30.8 -> 39 us; three
  // blocks per wave: 44.6 us)
  if (use_wide(C, ntile) && aligned16(w) && (ntile <= 256 || g_tok_path == 2)) {
    const long waves = ntile * (N / 32);
    const dim3 wgrid((unsigned)((waves + 3) / 4));
    hipStream_t s = as_stream(st);
    if (C == 64) hipLaunchKernelGGL((tok_head_fwd_wide<64, 1>), wgrid, dim3(256), 0, s, x, gamma, beta, w, bias, h_out, stats_out, qkv, B, P, eps);
    else         hipLaunchKernelGGL((tok_head_fwd_wide<128, 1>), wgrid, dim3(256), 0, s, x, gamma, beta, w, bias, h_out, stats_out, qkv, B, P, eps);
    return check_launch("afd_tok_head_fwd");
  }
  int npw = 32;
  for (int cand : {N, 192, 128, 96, 64, 32}) {
    if (cand > N || N % cand != 0) continue;
    if (sizeof(float) * ((size_t)cand * (C + 5) + 2 * C) > 64 * 1024) continue;
    npw = cand;
    if (wg_tiles * (N / cand) >= 256) break;
  }
  const size_t lds = sizeof(float) * ((size_t)npw * (C + 5) + 2 * C);
  const dim3 grid((unsigned)std::min<long>(wg_tiles, g_tok_max_wg > 0 ? g_tok_max_wg : 1024), (unsigned)(N / npw));
  hipStream_t s = as_stream(st);
  switch (C) {
    case 32:  hipLaunchKernelGGL(tok_head_fwd<32>, grid, dim3(256), lds, s, x, gamma, beta, w, bias, h_out, stats_out, qkv, B, N, P, eps, npw); break;
    case 64:  hipLaunchKernelGGL(tok_head_fwd<64>, grid, dim3(256), lds, s, x, gamma, beta, w, bias, h_out, stats_out, qkv, B, N, P, eps, npw); break;
    default:  hipLaunchKernelGGL(tok_head_fwd<128>, grid, dim3(256), lds, s, x, gamma, beta, w, bias, h_out, stats_out, qkv, B, N, P, eps, npw); break;
  }
  return check_launch("afd_tok_head_fwd");
}

int afd_tok_tail_fwd(const float* att, const float* x, const float* wo, const float* bo, const float* gamma, const float* beta,
                     const float* w1, const float* b1, const float* w2, const float* b2, float* a_out, float* stats_out,
                     float* f_out, float* u_out, float* g_out, float* out, int B, int C, int P, float eps, afd_stream_t st) {
  AFD_REQUIRE(att && x && wo && bo && gamma && beta && w1 && b1 && w2 && b2 && out && B > 0 && P > 0,
              "afd_tok_tail_fwd: NULL pointer or empty shape");
  AFD_REQUIRE(tok_ok(C), "afd_tok_tail_fwd: C=%d not covered (32, 64, 128)", C);
  AFD_REQUIRE((long)B * P * C < (1L << 31), "afd_tok_tail_fwd: tensor too large for 32-bit offsets");
  const bool train = a_out != nullptr;
  AFD_REQUIRE(!train || (stats_out && f_out && u_out && g_out), "afd_tok_tail_fwd: a_out given but a saved-tensor pointer is NULL");
  const long ntile = ((long)B * P + 31) / 32;
  hipStream_t s = as_stream(st);
  if (use_wide(C, ntile) && aligned16(wo) && aligned16(w1) && aligned16(w2)) {
    const dim3 wgrid((unsigned)((ntile + Wide<64>::TPW - 1) / Wide<64>::TPW)), wgrid128((unsigned)ntile);
#define AFD_TAILW(C_, T_, G_) hipLaunchKernelGGL((tok_tail_fwd_wide<C_, T_>), G_, dim3(256), 0, s, att, x, wo, bo, gamma, beta, w1, b1, w2, b2, \
                                                  a_out, stats_out, f_out, u_out, g_out, out, B, P, eps)
    if (C == 64) { if (train) AFD_TAILW(64, true, wgrid); else AFD_TAILW(64, false, wgrid); }
    else         { if (train) AFD_TAILW(128, true, wgrid128); else AFD_TAILW(128, false, wgrid128); }
#undef AFD_TAILW
    return check_launch("afd_tok_tail_fwd");
  }
  const size_t lds = chain_lds(C, 5);
  const dim3 grid(chain_grid(C, ntile));
#define AFD_TAIL(C_, T_) do { if (int rc = set_lds(tok_tail_fwd<C_, T_>, lds)) return rc; \
    hipLaunchKernelGGL((tok_tail_fwd<C_, T_>), grid, dim3(256), lds, s, att, x, wo, bo, gamma, beta, w1, b1, w2, b2, a_out, stats_out, \
                       f_out, u_out, g_out, out, B, P, eps); } while (0)
  switch (C) {
    case 32:  if (train) AFD_TAIL(32, true); else AFD_TAIL(32, false); break;
    case 64:  if (train) AFD_TAIL(64, true); else AFD_TAIL(64, false); break;
    default:  if (train) AFD_TAIL(128, true); else AFD_TAIL(128, false); break;
  }
#undef AFD_TAIL
  return check_launch("afd_tok_tail_fwd");
}

int afd_tok_tail_bwd(const float* d_out, const float* u, const float* a, const float* stats, const float* gamma, const float* w2,
                     const float* w1, const float* wo, float* du_out, float* df_out, float* da_out, float* datt_out, int B, int C,
                     int P, afd_stream_t st) {
  AFD_REQUIRE(d_out && u && a && stats && gamma && w2 && w1 && wo && du_out && df_out && da_out && datt_out && B > 0 && P > 0,
              "afd_tok_tail_bwd: NULL pointer or empty shape");
  AFD_REQUIRE(tok_ok(C), "afd_tok_tail_bwd: C=%d not covered (32, 64, 128)", C);
  AFD_REQUIRE((long)B * P * C < (1L << 31), "afd_tok_tail_bwd: tensor too large for 32-bit offsets");
  const long ntile = ((long)B * P + 31) / 32;
  hipStream_t s = as_stream(st);
  if (use_wide(C, ntile)) {
    if (C == 64) hipLaunchKernelGGL(tok_tail_bwd_wide<64>, dim3((unsigned)((ntile + 1) / 2)), dim3(256), 0, s, d_out, u, a, stats, gamma, w2, w1, wo, du_out, df_out, da_out, datt_out, B, P);
    else         hipLaunchKernelGGL(tok_tail_bwd_wide<128>, dim3((unsigned)ntile), dim3(256), 0, s, d_out, u, a, stats, gamma, w2, w1, wo, du_out, df_out, da_out, datt_out, B, P);
    return check_launch("afd_tok_tail_bwd");
  }
  const size_t lds = chain_lds(C, 1);
  const dim3 grid(chain_grid(C, ntile));
#define AFD_TB(C_) do { if (int rc = set_lds(tok_tail_bwd<C_>, lds)) return rc; \
    hipLaunchKernelGGL(tok_tail_bwd<C_>, grid, dim3(256), lds, s, d_out, u, a, stats, gamma, w2, w1, wo, du_out, df_out, da_out, datt_out, B, P); } while (0)
  switch (C) { case 32: AFD_TB(32); break; case 64: AFD_TB(64); break; default: AFD_TB(128); break; }
#undef AFD_TB
  return check_launch("afd_tok_tail_bwd");
}

int afd_tok_head_bwd(const float* dqkv, const float* x, const float* stats, const float* gamma, const float* w_in,
                     const float* d_res, float* dh_out, float* dx_out, int B, int C, int P, afd_stream_t st) {
  AFD_REQUIRE(dqkv && x && stats && gamma && w_in && d_res && dx_out && B > 0 && P > 0, "afd_tok_head_bwd: NULL pointer or empty shape");
  AFD_REQUIRE(tok_ok(C), "afd_tok_head_bwd: C=%d not covered (32, 64, 128)", C);
  AFD_REQUIRE((long)B * P * 3 * C < (1L << 31), "afd_tok_head_bwd: tensor too large for 32-bit offsets");
  const long ntile = ((long)B * P + 31) / 32;
  hipStream_t s = as_stream(st);
  if (use_wide(C, ntile)) {
    if (C == 64) hipLaunchKernelGGL(tok_head_bwd_wide<64>, dim3((unsigned)((ntile + 1) / 2)), dim3(256), 0, s, dqkv, x, stats, gamma, w_in, d_res, dh_out, dx_out, B, P);
    else         hipLaunchKernelGGL(tok_head_bwd_wide<128>, dim3((unsigned)ntile), dim3(256), 0, s, dqkv, x, stats, gamma, w_in, d_res, dh_out, dx_out, B, P);
    return check_launch("afd_tok_head_bwd");
  }
  const size_t lds = chain_lds(C, 1);
  const dim3 grid(chain_grid(C, ntile));
#define AFD_HB(C_) do { if (int rc = set_lds(tok_head_bwd<C_>, lds)) return rc; \
    hipLaunchKernelGGL(tok_head_bwd<C_>, grid, dim3(256), lds, s, dqkv, x, stats, gamma, w_in, d_res, dh_out, dx_out, B, P); } while (0)
  switch (C) { case 32: AFD_HB(32); break; case 64: AFD_HB(64); break; default: AFD_HB(128); break; }
#undef AFD_HB
  return check_launch("afd_tok_head_bwd");
}

}  // extern "C"

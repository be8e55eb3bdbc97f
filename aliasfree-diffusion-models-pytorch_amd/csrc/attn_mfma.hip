// attn_mfma.hip -- attention core for head dim 8 (the 16x16 and 32x32 maps: 70 % of attention FLOPs) with
// the two d-contractions of every pass on the fp32 matrix cores.
//
// S^T = K Q^T (and dP^T = V dO^T) are matrix-core products over d = 8 -- at fp32 accuracy on the bf16 path (three
// v_mfma_f32_32x32x16_bf16 per 32x32 tile on exact three-piece splits of the operands, see split3 below).  Their accumulator layout -- lane = query column (l & 31), the 16 registers = 16 of the 32 keys, the other 16 in
// lane ^ 32 -- is kept for everything that follows: the online softmax is a max over registers plus one
// cross-half exchange, and the rank-8 products (P V, dS K, P^T dO, dS^T Q) are vector FMAs against LDS rows
// that both tiles of a wave share.  Compared with the all-VALU kernels this removes 8 of 21 (forward),
// 16 of 30 (dQ) and 16 of 38 (dK/dV) vector instructions per (query, key) pair.
// Scores live in the log2 domain: log2(e)/sqrt(d) is folded into the Q (K) fragments, exp is v_exp_f32.
// Requirements: d == 8, L % 256 == 0 (others use attn.hip).  Deterministic, no atomics.
#include "common.h"

namespace afd {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kD = 8, kTK = 64;                      // head dim, rows of the streamed operand per LDS tile
constexpr int kNE = kD * kTK / 256;                  // staged values per thread and operand
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, kWave); }     // value of lane ^ 32
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// stage a (8, L) j-major operand tile [r0, r0+64) into LDS both d-major [8][64] and row-major [64][8]
__device__ __forceinline__ void stage_both(const float* __restrict__ src, int L, int r0, float* __restrict__ dmaj,
                                           float* __restrict__ rmaj, float mul) {
  for (int i = threadIdx.x; i < kD * kTK; i += 256) {
    const int j = i / kTK, rr = i % kTK;
    const float v = src[(long)j * L + r0 + rr] * mul;
    dmaj[j * kTK + rr] = v;
    if (rmaj) rmaj[rr * kD + j] = v;
  }
}
// rank-8 updates run as packed f32 FMAs (v_pk_fma_f32, the scalar broadcast through op_sel): measured 5.4 cycles per
// wave instruction against 4.3 for one v_fma_f32, i.e. 1.6x the FMA rate.  An LDS row of 8 is four register pairs.
using f2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ void load_row8(const float* __restrict__ p, f2 (&v)[4]) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = (f2){a.x, a.y}; v[1] = (f2){a.z, a.w}; v[2] = (f2){b.x, b.y}; v[3] = (f2){b.z, b.w};
}
__device__ __forceinline__ void axpy8(f2 (&acc)[4], float s, const f2 (&v)[4]) {
  const f2 ss = {s, s};
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = __builtin_elementwise_fma(ss, v[i], acc[i]);
}

// ---- fp32-accurate d-contractions on the bf16 matrix path ----------------------------------------------------------
// x = x1 + x2 + x3 exactly (three bf16 pieces, 8 mantissa bits each); the six leading cross terms of a product of two
// such sums (a1b1, a2b1, a1b2, a3b1, a2b2, a1b3) carry it to 2^-24.  Over d = 8 that is 48 bf16 products per (row,
// column) pair = THREE 32x32x16 MFMAs whose two lane halves (k = 8 half + d) hold two different terms, against four
// 32x32x2 fp32 MFMAs.  Measured on MI355X (tools/micro/bf16x3.hip): max error 1.6e-7 of sum|a b| (fp32 MFMA: 1.2e-7)
// at 0.58x the matrix-pipe time.  The accumulator layout is the fp32 one.
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
using bf2 = __attribute__((ext_vector_type(2))) __bf16;
__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x; const float r = x - (float)a;
  b = (__bf16)r; const float r2 = r - (float)b;
  c = (__bf16)r2;
}
// B side (the row a lane keeps in registers for the whole kernel): halves (0 | 1) carry  M1: b1 | b1,  M2: b2 | b1,  M3: b2 | b3
__device__ __forceinline__ void row_frags(const float (&x)[kD], int half, bf8 (&f)[3]) {
  bf8 p1, p2, p3;
#pragma unroll
  for (int d = 0; d < kD; ++d) { __bf16 a, b, c; split3(x[d], a, b, c); p1[d] = a; p2[d] = b; p3[d] = c; }
  f[0] = p1; f[1] = half ? p1 : p2; f[2] = half ? p3 : p2;
}
// A side (the streamed operand): pieces in LDS as [piece][row][8 bf16] (one ds_read_b128 per fragment);
// halves carry  M1: a1 | a2,  M2: a1 | a3,  M3: a2 | a1.  A thread stages the d-pair (2 jp, 2 jp + 1) of row rr.
__device__ __forceinline__ void stage_pieces(uint32_t* __restrict__ P, int rr, int jp, float v0, float v1) {
  __bf16 a0, b0, c0, a1, b1, c1;
  split3(v0, a0, b0, c0); split3(v1, a1, b1, c1);
  P[(0 * kTK + rr) * 4 + jp] = __builtin_bit_cast(uint32_t, (bf2){a0, a1});
  P[(1 * kTK + rr) * 4 + jp] = __builtin_bit_cast(uint32_t, (bf2){b0, b1});
  P[(2 * kTK + rr) * 4 + jp] = __builtin_bit_cast(uint32_t, (bf2){c0, c1});
}
__device__ __forceinline__ void load_frags(const uint32_t* __restrict__ P, int row, int half, bf8 (&a)[3]) {
  const bf8* p = reinterpret_cast<const bf8*>(P);
  a[0] = p[(half ? 1 : 0) * kTK + row]; a[1] = p[(half ? 2 : 0) * kTK + row]; a[2] = p[(half ? 0 : 1) * kTK + row];
}
__device__ __forceinline__ f32x16 dot8x3(const bf8 (&a)[3], const bf8 (&b)[3]) {
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
  for (int m = 0; m < 3; ++m) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[m], c, 0, 0, 0);
  return c;
}

// ------------------------------------------------------------------------------------------------
// forward: workgroup = 256 queries (4 waves x 2 query tiles of 32), streams K/V in tiles of 64 keys
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_fwd_mfma8(const float* __restrict__ qkv, float* __restrict__ o,
                                                      float* __restrict__ lse, int heads, int L, float scale) {
  __shared__ __attribute__((aligned(16))) uint32_t Kp[3 * kTK * 4];  // K pieces (MFMA A fragments)
  __shared__ __attribute__((aligned(16))) float Vr[kTK * kD];       // V, row-major (P V rows)
  const int b = blockIdx.z, h = blockIdx.y, C = heads * kD;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const float* qp = qkv + ((long)b * 3 * C + h * kD) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const int q0 = blockIdx.x * 256 + wv * 64;                          // this wave's first query
  bf8 bq[2][3];                                                       // B fragments of the two query tiles
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float x[kD];
#pragma unroll
    for (int d = 0; d < kD; ++d) x[d] = qp[(long)d * L + q0 + j * 32 + l31] * (scale * kLog2e);
    row_frags(x, half, bq[j]);
  }
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  f2 oa[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) oa[j][i] = (f2){0.f, 0.f};

  // the next K / V tile travels global -> registers while the current one is multiplied: a thread owns the d-pair
  // (2 jp, 2 jp + 1) of row rr (kNE = 2 values per operand)
  static_assert(kNE == 2, "staging assumes one d-pair per thread");
  const int jp = threadIdx.x >> 6, rr = threadIdx.x & 63;
  float kreg[kNE], vreg[kNE];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < kNE; ++e) {
      kreg[e] = kp[(long)(2 * jp + e) * L + k0 + rr];
      vreg[e] = vp[(long)(2 * jp + e) * L + k0 + rr];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < L; k0 += kTK) {
    __syncthreads();
    stage_pieces(Kp, rr, jp, kreg[0], kreg[1]);
    *reinterpret_cast<float2*>(Vr + rr * kD + 2 * jp) = make_float2(vreg[0], vreg[1]);
    __syncthreads();
    if (k0 + kTK < L) fetch(k0 + kTK);
#pragma unroll
    for (int kt = 0; kt < kTK / 32; ++kt) {
      bf8 ak[3];
      load_frags(Kp, kt * 32 + l31, half, ak);
      f32x16 sc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        sc[j] = dot8x3(ak, bq[j]);
        float mx = sc[j][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[j][r]);
        mx = fmaxf(mx, xhalf(mx));
        const float mn = fmaxf(m[j], mx);
        const float alpha = __builtin_amdgcn_exp2f(m[j] - mn);        // m = -inf first: exp2(-inf) = 0
        l[j] *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i) oa[j][i] *= alpha;
        m[j] = mn;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f2 v[4];
        load_row8(Vr + (kt * 32 + acc_row(r, half)) * kD, v);         // shared by both query tiles
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float p = __builtin_amdgcn_exp2f(sc[j][r] - m[j]);
          l[j] += p;
          axpy8(oa[j], p, v);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float lt = l[j] + xhalf(l[j]);
    const float inv = 1.0f / lt;
    const int qi = q0 + j * 32 + l31;
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      const float mine = oa[j][d >> 1][d & 1];
      const float t = mine + xhalf(mine);
      if (half == 0) o[((long)b * C + h * kD + d) * L + qi] = t * inv;
    }
    if (half == 0) lse[((long)b * heads + h) * L + qi] = (m[j] + __builtin_amdgcn_logf(lt)) * kLn2;   // v_log_f32 = log2
  }
}

// ------------------------------------------------------------------------------------------------
// dQ (and delta = rowsum(dO * O)): same tiling as forward; S^T and dP^T on MFMA
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_mfma8(const float* __restrict__ qkv, const float* __restrict__ o,
                                                         const float* __restrict__ d_o, const float* __restrict__ lse,
                                                         float* __restrict__ dqkv, float* __restrict__ delta_out,
                                                         int heads, int L, float scale) {
  __shared__ __attribute__((aligned(16))) uint32_t Kp[3 * kTK * 4];  // K pieces (A fragments of S^T)
  __shared__ __attribute__((aligned(16))) uint32_t Vp[3 * kTK * 4];  // V pieces (A fragments of dP^T)
  __shared__ __attribute__((aligned(16))) float Kr[kTK * kD];
  const int b = blockIdx.z, h = blockIdx.y, C = heads * kD;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const float* qp = qkv + ((long)b * 3 * C + h * kD) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const long ob = ((long)b * C + h * kD) * L;
  const int q0 = blockIdx.x * 256 + wv * 64;
  bf8 bq[2][3], bg[2][3];
  float lsq[2], dlt[2];
  f2 dq[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qi = q0 + j * 32 + l31;
    float dpart = 0.f, xq[kD], xg[kD];
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      xq[d] = qp[(long)d * L + qi] * (scale * kLog2e);
      xg[d] = d_o[ob + (long)d * L + qi];
      dpart = fmaf(xg[d], o[ob + (long)d * L + qi], dpart);
    }
    row_frags(xq, half, bq[j]);
    row_frags(xg, half, bg[j]);
    dlt[j] = dpart;
    lsq[j] = lse[((long)b * heads + h) * L + qi] * kLog2e;
    if (half == 0) delta_out[((long)b * heads + h) * L + qi] = dlt[j];
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[j][i] = (f2){0.f, 0.f};
  }
  const int jp = threadIdx.x >> 6, rr = threadIdx.x & 63;      // staging: d-pair (2 jp, 2 jp + 1) of row rr
  float kreg[kNE], vreg[kNE];                           // next K / V tile in flight during the multiplies
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < kNE; ++e) {
      kreg[e] = kp[(long)(2 * jp + e) * L + k0 + rr];
      vreg[e] = vp[(long)(2 * jp + e) * L + k0 + rr];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < L; k0 += kTK) {
    __syncthreads();
    stage_pieces(Kp, rr, jp, kreg[0], kreg[1]);
    stage_pieces(Vp, rr, jp, vreg[0], vreg[1]);
    *reinterpret_cast<float2*>(Kr + rr * kD + 2 * jp) = make_float2(kreg[0], kreg[1]);
    __syncthreads();
    if (k0 + kTK < L) fetch(k0 + kTK);
#pragma unroll
    for (int kt = 0; kt < kTK / 32; ++kt) {
      bf8 ak[3], av[3];
      load_frags(Kp, kt * 32 + l31, half, ak);
      load_frags(Vp, kt * 32 + l31, half, av);
      f32x16 sc[2], dp[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        sc[j] = dot8x3(ak, bq[j]);
        dp[j] = dot8x3(av, bg[j]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f2 kr[4];
        load_row8(Kr + (kt * 32 + acc_row(r, half)) * kD, kr);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float ds = __builtin_amdgcn_exp2f(sc[j][r] - lsq[j]) * (dp[j][r] - dlt[j]);
          axpy8(dq[j], ds, kr);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qi = q0 + j * 32 + l31;
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      const float mine = dq[j][d >> 1][d & 1];
      const float t = mine + xhalf(mine);
      if (half == 0) dqkv[((long)b * 3 * C + h * kD + d) * L + qi] = t * scale;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: workgroup = 256 keys (4 waves x 2 key tiles), streams Q / dO / lse / delta in tiles of 64 queries.
// Tiles are [query rows x key columns]: lane = key, registers = 16 of 32 queries.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_mfma8(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ dqkv, int heads, int L, float scale) {
  __shared__ __attribute__((aligned(16))) uint32_t Qp[3 * kTK * 4];  // Q pieces  (A fragments of S)
  __shared__ __attribute__((aligned(16))) uint32_t Gp[3 * kTK * 4];  // dO pieces (A fragments of dP)
  __shared__ __attribute__((aligned(16))) float Qr[kTK * kD];       // rows for dK += dS^T Q
  __shared__ __attribute__((aligned(16))) float Gr[kTK * kD];       // rows for dV += P^T dO
  __shared__ float Ls[kTK];
  __shared__ float Ds[kTK];
  const int b = blockIdx.z, h = blockIdx.y, C = heads * kD;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const float* qp = qkv + ((long)b * 3 * C + h * kD) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const float* gp = d_o + ((long)b * C + h * kD) * L;
  const float* lp = lse + ((long)b * heads + h) * L;
  const float* dlp = delta + ((long)b * heads + h) * L;
  const int key0 = blockIdx.x * 256 + wv * 64;
  // B fragments of the wave's two key tiles (K scaled, V): a wave-private LDS image, one ds_read_b128 per fragment and
  // lane (in registers they cost 48 VGPRs, and the kernel must keep 2 waves / SIMD)
  __shared__ bf8 Bf[4][2][2][3][64];
  f2 dk[2][4], dv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float xk[kD], xv[kD];
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      xk[d] = kp[(long)d * L + key0 + j * 32 + l31] * (scale * kLog2e);
      xv[d] = vp[(long)d * L + key0 + j * 32 + l31];
    }
    bf8 t[3];
    row_frags(xk, half, t);
#pragma unroll
    for (int m = 0; m < 3; ++m) Bf[wv][j][0][m][lane] = t[m];
    row_frags(xv, half, t);
#pragma unroll
    for (int m = 0; m < 3; ++m) Bf[wv][j][1][m][lane] = t[m];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dk[j][i] = (f2){0.f, 0.f}; dv[j][i] = (f2){0.f, 0.f}; }
  }
  float qreg[kNE], greg[kNE], lreg = 0.f, dreg = 0.f;       // next Q / dO / lse / delta tile in flight during the multiplies
  const int jp = threadIdx.x >> 6, rr = threadIdx.x & 63;      // staging: d-pair (2 jp, 2 jp + 1) of row rr
  auto fetch = [&](int t0) {
#pragma unroll
    for (int e = 0; e < kNE; ++e) {
      qreg[e] = qp[(long)(2 * jp + e) * L + t0 + rr];
      greg[e] = gp[(long)(2 * jp + e) * L + t0 + rr];
    }
    if (threadIdx.x < kTK) { lreg = lp[t0 + threadIdx.x] * kLog2e; dreg = dlp[t0 + threadIdx.x]; }
  };
  fetch(0);
  for (int t0 = 0; t0 < L; t0 += kTK) {
    __syncthreads();
    stage_pieces(Qp, rr, jp, qreg[0], qreg[1]);
    stage_pieces(Gp, rr, jp, greg[0], greg[1]);
    *reinterpret_cast<float2*>(Qr + rr * kD + 2 * jp) = make_float2(qreg[0], qreg[1]);
    *reinterpret_cast<float2*>(Gr + rr * kD + 2 * jp) = make_float2(greg[0], greg[1]);
    if (threadIdx.x < kTK) { Ls[threadIdx.x] = lreg; Ds[threadIdx.x] = dreg; }
    __syncthreads();
    if (t0 + kTK < L) fetch(t0 + kTK);
#pragma unroll
    for (int qt = 0; qt < kTK / 32; ++qt) {
      bf8 aq[3], ag[3];
      load_frags(Qp, qt * 32 + l31, half, aq);
      load_frags(Gp, qt * 32 + l31, half, ag);
      f32x16 sc[2], dp[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf8 bkj[3], bvj[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) { bkj[m] = Bf[wv][j][0][m][lane]; bvj[m] = Bf[wv][j][1][m][lane]; }
        sc[j] = dot8x3(aq, bkj);
        dp[j] = dot8x3(ag, bvj);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = qt * 32 + acc_row(r, half);
        f2 qrow[4], grow[4];
        load_row8(Qr + qr * kD, qrow);
        load_row8(Gr + qr * kD, grow);
        const float ls = Ls[qr], dl = Ds[qr];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float p = __builtin_amdgcn_exp2f(sc[j][r] - ls);
          const float ds = p * (dp[j][r] - dl);
          axpy8(dv[j], p, grow);
          axpy8(dk[j], ds, qrow);
        }
        if ((r & 1) == 1) asm volatile("" ::: "memory");      // bound how many LDS rows are in flight (register budget)
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ki = key0 + j * 32 + l31;
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      const float mk = dk[j][d >> 1][d & 1], mv = dv[j][d >> 1][d & 1];
      const float tk = mk + xhalf(mk), tv = mv + xhalf(mv);
      if (half == 0) {
        dqkv[((long)b * 3 * C + C + h * kD + d) * L + ki] = tk * scale;
        dqkv[((long)b * 3 * C + 2 * C + h * kD + d) * L + ki] = tv;
      }
    }
  }
}

// A one-pass backward (S, dP, the exponentials and dS computed once, P and dS transposed through wave-private LDS so that
// dK / dV become in-lane sums) was built and measured in round 1: register-bound (the loop-invariant Q / dO rows take
// ~250 registers, or spill), 2.7-13x slower than the two passes above.  It was removed in round 2; the numbers are in
// DESIGN.md section 6.

// host-side launchers used by attn.hip
bool attn_mfma8_ok(int d, int L) { return d == 8 && L % 256 == 0; }
void attn_mfma8_fwd(const float* qkv, float* o, float* lse, int B, int heads, int L, float sc, hipStream_t s) {
  hipLaunchKernelGGL(attn_fwd_mfma8, dim3(L / 256, heads, B), dim3(256), 0, s, qkv, o, lse, heads, L, sc);
}
void attn_mfma8_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, float* delta,
                    int B, int heads, int L, float sc, hipStream_t s) {
  hipLaunchKernelGGL(attn_bwd_dq_mfma8, dim3(L / 256, heads, B), dim3(256), 0, s, qkv, o, d_o, lse, dqkv, delta, heads, L, sc);
  hipLaunchKernelGGL(attn_bwd_dkv_mfma8, dim3(L / 256, heads, B), dim3(256), 0, s, qkv, d_o, lse, delta, dqkv, heads, L, sc);
}

}  // namespace afd

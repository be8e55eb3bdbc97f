// attn_mfma.hip -- attention core for head dim 8 (the 16x16 and 32x32 maps: 70 % of attention FLOPs) with
// the two d-contractions of every pass on the fp32 matrix cores.
//
// S^T = K Q^T (and dP^T = V dO^T) are 32x32x2 MFMAs over d = 8 (4 instructions per 32x32 tile).  Their
// accumulator layout -- lane = query column (l & 31), the 16 registers = 16 of the 32 keys, the other 16 in
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

// ------------------------------------------------------------------------------------------------
// forward: workgroup = 256 queries (4 waves x 2 query tiles of 32), streams K/V in tiles of 64 keys
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_fwd_mfma8(const float* __restrict__ qkv, float* __restrict__ o,
                                                      float* __restrict__ lse, int heads, int L, float scale) {
  __shared__ __attribute__((aligned(16))) float Kd[kD * kTK];       // K, d-major  (MFMA A fragments)
  __shared__ __attribute__((aligned(16))) float Vr[kTK * kD];       // V, row-major (P V rows)
  const int b = blockIdx.z, h = blockIdx.y, C = heads * kD;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const float* qp = qkv + ((long)b * 3 * C + h * kD) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const int q0 = blockIdx.x * 256 + wv * 64;                          // this wave's first query
  float bq[2][4];                                                     // B fragments of the two query tiles
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) bq[j][s] = qp[(long)(2 * s + half) * L + q0 + j * 32 + l31] * (scale * kLog2e);
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  f2 oa[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) oa[j][i] = (f2){0.f, 0.f};

  // the next K / V tile travels global -> registers while the current one is multiplied (2 x 2 values per thread)
  float kreg[kNE], vreg[kNE];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < kNE; ++e) {
      const int i = threadIdx.x + 256 * e, j = i / kTK, rr = i % kTK;
      kreg[e] = kp[(long)j * L + k0 + rr];
      vreg[e] = vp[(long)j * L + k0 + rr];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < L; k0 += kTK) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kNE; ++e) {
      const int i = threadIdx.x + 256 * e, j = i / kTK, rr = i % kTK;
      Kd[j * kTK + rr] = kreg[e];
      Vr[rr * kD + j] = vreg[e];
    }
    __syncthreads();
    if (k0 + kTK < L) fetch(k0 + kTK);
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
#pragma unroll
        for (int s = 0; s < 4; ++s) sc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[s], bq[j][s], sc[j], 0, 0, 0);
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
  __shared__ __attribute__((aligned(16))) float Kd[kD * kTK];
  __shared__ __attribute__((aligned(16))) float Vd[kD * kTK];
  __shared__ __attribute__((aligned(16))) float Kr[kTK * kD];
  const int b = blockIdx.z, h = blockIdx.y, C = heads * kD;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const float* qp = qkv + ((long)b * 3 * C + h * kD) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const long ob = ((long)b * C + h * kD) * L;
  const int q0 = blockIdx.x * 256 + wv * 64;
  float bq[2][4], bg[2][4], lsq[2], dlt[2];
  f2 dq[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qi = q0 + j * 32 + l31;
    float dpart = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int d = 2 * s + half;
      bq[j][s] = qp[(long)d * L + qi] * (scale * kLog2e);
      bg[j][s] = d_o[ob + (long)d * L + qi];
      dpart = fmaf(bg[j][s], o[ob + (long)d * L + qi], dpart);
    }
    dlt[j] = dpart + xhalf(dpart);
    lsq[j] = lse[((long)b * heads + h) * L + qi] * kLog2e;
    if (half == 0) delta_out[((long)b * heads + h) * L + qi] = dlt[j];
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[j][i] = (f2){0.f, 0.f};
  }
  float kreg[kNE], vreg[kNE];                           // next K / V tile in flight during the multiplies
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < kNE; ++e) {
      const int i = threadIdx.x + 256 * e, j = i / kTK, rr = i % kTK;
      kreg[e] = kp[(long)j * L + k0 + rr];
      vreg[e] = vp[(long)j * L + k0 + rr];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < L; k0 += kTK) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kNE; ++e) {
      const int i = threadIdx.x + 256 * e, j = i / kTK, rr = i % kTK;
      Kd[j * kTK + rr] = kreg[e]; Kr[rr * kD + j] = kreg[e];
      Vd[j * kTK + rr] = vreg[e];
    }
    __syncthreads();
    if (k0 + kTK < L) fetch(k0 + kTK);
#pragma unroll
    for (int kt = 0; kt < kTK / 32; ++kt) {
      float ak[4], av[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        ak[s] = Kd[(2 * s + half) * kTK + kt * 32 + l31];
        av[s] = Vd[(2 * s + half) * kTK + kt * 32 + l31];
      }
      f32x16 sc[2], dp[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[j][r] = 0.f; dp[j][r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          sc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[s], bq[j][s], sc[j], 0, 0, 0);
          dp[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bg[j][s], dp[j], 0, 0, 0);
        }
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
  __shared__ __attribute__((aligned(16))) float Qd[kD * kTK];       // Q, d-major  (A fragments of S)
  __shared__ __attribute__((aligned(16))) float Gd[kD * kTK];       // dO, d-major (A fragments of dP)
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
  float bk[2][4], bv[2][4];
  f2 dk[2][4], dv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bk[j][s] = kp[(long)(2 * s + half) * L + key0 + j * 32 + l31] * (scale * kLog2e);
      bv[j][s] = vp[(long)(2 * s + half) * L + key0 + j * 32 + l31];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { dk[j][i] = (f2){0.f, 0.f}; dv[j][i] = (f2){0.f, 0.f}; }
  }
  float qreg[kNE], greg[kNE], lreg = 0.f, dreg = 0.f;       // next Q / dO / lse / delta tile in flight during the multiplies
  auto fetch = [&](int t0) {
#pragma unroll
    for (int e = 0; e < kNE; ++e) {
      const int i = threadIdx.x + 256 * e, j = i / kTK, rr = i % kTK;
      qreg[e] = qp[(long)j * L + t0 + rr];
      greg[e] = gp[(long)j * L + t0 + rr];
    }
    if (threadIdx.x < kTK) { lreg = lp[t0 + threadIdx.x] * kLog2e; dreg = dlp[t0 + threadIdx.x]; }
  };
  fetch(0);
  for (int t0 = 0; t0 < L; t0 += kTK) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kNE; ++e) {
      const int i = threadIdx.x + 256 * e, j = i / kTK, rr = i % kTK;
      Qd[j * kTK + rr] = qreg[e]; Qr[rr * kD + j] = qreg[e];
      Gd[j * kTK + rr] = greg[e]; Gr[rr * kD + j] = greg[e];
    }
    if (threadIdx.x < kTK) { Ls[threadIdx.x] = lreg; Ds[threadIdx.x] = dreg; }
    __syncthreads();
    if (t0 + kTK < L) fetch(t0 + kTK);
#pragma unroll
    for (int qt = 0; qt < kTK / 32; ++qt) {
      float aq[4], ag[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        aq[s] = Qd[(2 * s + half) * kTK + qt * 32 + l31];
        ag[s] = Gd[(2 * s + half) * kTK + qt * 32 + l31];
      }
      f32x16 sc[2], dp[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[j][r] = 0.f; dp[j][r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          sc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[s], bk[j][s], sc[j], 0, 0, 0);
          dp[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[s], bv[j][s], dp[j], 0, 0, 0);
        }
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

// ------------------------------------------------------------------------------------------------
// fused backward (dQ, dK, dV in ONE kernel; S, dP, the exponentials and dS are computed once instead of twice).
// Workgroup = one (batch, head); wave w owns queries [128 w, 128 w + 128) as 4 tiles of 32.  The workgroup makes four
// passes (one query tile per wave per pass: the tile's Q / dO fragments, lse, delta and its dQ accumulators are the
// only per-tile state, all in registers) and in each pass walks the keys in blocks of 32.  Per (key block, tile):
// S^T and dP^T on the MFMA (lane = query, registers = keys), P and dS in that layout, dQ += dS K in-lane.
// dK / dV need the sum over QUERIES, i.e. over the lane index: the P and dS tiles go through a wave-private LDS image
// (4 ds_write_b128 + 16 ds_read_b32 each, no barrier) and come back with lane = key, where dV += P^T dO and
// dK += dS^T Q are in-lane rank-8 updates against broadcast LDS rows.  Each key block ends with a fixed-order sum of
// the waves' partials; passes 1..3 add to what the earlier passes wrote (fixed order: deterministic, no atomics).
// ------------------------------------------------------------------------------------------------
constexpr int kTS = 36;                              // row stride of the transposition images (conflict-free b128 writes)
constexpr int kFusedWaveFloats = 2 * 32 * kTS + 2 * 32 * kD;

template <int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_fused8(const float* __restrict__ qkv, const float* __restrict__ o,
                                                            const float* __restrict__ d_o, const float* __restrict__ lse,
                                                            float* __restrict__ dqkv, float* __restrict__ delta_out,
                                                            int heads, float scale) {
  constexpr int L = 128 * NW, NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Kd = smem;                                  // [8][32]  K block, d-major (A fragments of S^T)
  float* Vd = Kd + kD * 32;                          // [8][32]  V block, d-major (A fragments of dP^T)
  float* Kr = Vd + kD * 32;                          // [32][8]  K rows (dQ updates)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  float* Pimg = Kr + 32 * kD + wv * kFusedWaveFloats;  // wave-private: [32 q][kTS] P tile
  float* Dimg = Pimg + 32 * kTS;                     //               [32 q][kTS] dS tile
  float* Qr = Dimg + 32 * kTS;                       //               [32 q][8] raw Q rows of the current tile
  float* Gr = Qr + 32 * kD;                          //               [32 q][8] dO rows
  const int b = blockIdx.y, h = blockIdx.x, C = heads * kD;
  const float* qp = qkv + ((long)b * 3 * C + h * kD) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const long ob = ((long)b * C + h * kD) * L;

#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    const int qi = wv * 128 + pass * 32 + l31;        // this lane's query in the pass's tile
    float bq[4], bg[4];
    float dpart = 0.f;
    __syncthreads();                                  // (the previous pass's reducers are done with the image areas)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int d = 2 * s + half;
      const float qv = qp[(long)d * L + qi], gv = d_o[ob + (long)d * L + qi];
      bq[s] = qv * (scale * kLog2e);
      bg[s] = gv;
      dpart = fmaf(gv, o[ob + (long)d * L + qi], dpart);
      Qr[l31 * kD + d] = qv;
      Gr[l31 * kD + d] = gv;
    }
    const float dlt = dpart + xhalf(dpart);
    const float lsq = lse[((long)b * heads + h) * L + qi] * kLog2e;
    if (half == 0) delta_out[((long)b * heads + h) * L + qi] = dlt;
    f2 dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[i] = (f2){0.f, 0.f};

#pragma unroll 1
    for (int k0 = 0; k0 < L; k0 += 32) {
      __syncthreads();                                // K / V block and the partial-sum areas are free again
      for (int i = threadIdx.x; i < 2 * kD * 32; i += NT) {
        const int which = i >> 8, e = i & 255, d = e >> 5, kk = e & 31;
        const float v = (which ? vp : kp)[(long)d * L + k0 + kk];
        if (which) Vd[d * 32 + kk] = v; else { Kd[d * 32 + kk] = v; Kr[kk * kD + d] = v; }
      }
      __syncthreads();
      // the tile's Q / dO rows never change inside a pass, so their loads are invariant in this loop -- and hoisting the
      // 256 values per lane out of it costs 256 registers: launder the base pointers once per key block
      const float* Qk = Qr; const float* Gk = Gr;
      asm volatile("" : "+v"(Qk), "+v"(Gk));
      f32x16 sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Kd[(2 * s + half) * 32 + l31], bq[s], sc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vd[(2 * s + half) * 32 + l31], bg[s], dp, 0, 0, 0);
      }
      // lane = query: P, dS, dQ += dS K; the tiles leave for the transposition images 4 keys at a time
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float pv[4], dv4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * g + i;
          f2 kr[4];
          load_row8(Kr + acc_row(r, half) * kD, kr);
          const float p = __builtin_amdgcn_exp2f(sc[r] - lsq);
          const float ds = p * (dp[r] - dlt);
          axpy8(dq, ds, kr);
          pv[i] = p; dv4[i] = ds;
        }
        *reinterpret_cast<float4*>(Pimg + l31 * kTS + 8 * g + 4 * half) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        *reinterpret_cast<float4*>(Dimg + l31 * kTS + 8 * g + 4 * half) = make_float4(dv4[0], dv4[1], dv4[2], dv4[3]);
        __builtin_amdgcn_sched_barrier(0);              // 4 keys at a time: front-loading all 16 LDS rows spills
      }
      // lane = key (l31), this half takes queries 16 half .. 16 half + 15 of the tile
      f2 dkp[4], dvp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { dkp[i] = (f2){0.f, 0.f}; dvp[i] = (f2){0.f, 0.f}; }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = 16 * half + r;
        const float p = Pimg[q * kTS + l31], ds = Dimg[q * kTS + l31];
        f2 grow[4], qrow[4];
        load_row8(Gk + q * kD, grow);
        load_row8(Qk + q * kD, qrow);
        axpy8(dvp, p, grow);
        axpy8(dkp, ds, qrow);
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      // this wave's partial dK / dV rows of the block -> its own image area [32 keys][16], then a fixed-order sum
#pragma unroll
      for (int d = 0; d < kD; ++d) {
        const float mk = dkp[d >> 1][d & 1], mv = dvp[d >> 1][d & 1];
        const float tk = mk + xhalf(mk), tv = mv + xhalf(mv);
        if (half == 0) { Pimg[l31 * 16 + d] = tk; Pimg[l31 * 16 + 8 + d] = tv; }
      }
      __syncthreads();
      for (int t = threadIdx.x; t < 32 * 16; t += NT) {
        const int kk = t & 31, c = t >> 5;            // consecutive threads = consecutive keys: 128-B stores
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += (Kr + 32 * kD + w * kFusedWaveFloats)[kk * 16 + c];
        const int chan = (c < 8) ? (C + h * kD + c) : (2 * C + h * kD + (c - 8));
        float* dst = dqkv + ((long)b * 3 * C + chan) * L + k0 + kk;
        const float val = (c < 8) ? sum * scale : sum;
        *dst = pass ? *dst + val : val;                 // passes add in order: deterministic
      }
    }
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      const float mine = dq[d >> 1][d & 1];
      const float t = mine + xhalf(mine);
      if (half == 0) dqkv[((long)b * 3 * C + h * kD + d) * L + qi] = t * scale;
    }
  }
}

bool attn_fused8_ok(int d, int L) { return d == 8 && (L == 256 || L == 512 || L == 1024); }
template <int NW>
static void attn_fused8_launch(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, float* delta,
                               int B, int heads, float sc, hipStream_t s) {
  const size_t lds = sizeof(float) * (3 * 32 * kD + (size_t)NW * kFusedWaveFloats);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused8<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_bwd_fused8<NW>), dim3(heads, B), dim3(64 * NW), lds, s, qkv, o, d_o, lse, dqkv, delta, heads, sc);
}
void attn_fused8_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, float* delta,
                     int B, int heads, int L, float sc, hipStream_t s) {
  if (L == 256) attn_fused8_launch<2>(qkv, o, d_o, lse, dqkv, delta, B, heads, sc, s);
  else if (L == 512) attn_fused8_launch<4>(qkv, o, d_o, lse, dqkv, delta, B, heads, sc, s);
  else attn_fused8_launch<8>(qkv, o, d_o, lse, dqkv, delta, B, heads, sc, s);
}

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

// attn_mfma.hip -- attention core for head dim 8 and 16 (the 16x16 and 32x32 maps: sa1, sa5, sa6 -- 90 % of attention FLOPs)
// with the two d-contractions of every pass on the matrix cores.  (Written for d = 8; templated on D in round 2: at d = 16
// a lane half carries 8 of the 16 head dims, so a 32x32x16 MFMA holds ONE cross term and a tile takes six.)
//
// S^T = K Q^T (and dP^T = V dO^T) are matrix-core products over d = 8 -- at fp32 accuracy on the bf16 path (three
// v_mfma_f32_32x32x16_bf16 per 32x32 tile on exact three-piece splits of the operands, see split3 below).  Their accumulator layout -- lane = query column (l & 31), the 16 registers = 16 of the 32 keys, the other 16 in
// lane ^ 32 -- is kept for everything that follows: the online softmax is a max over registers plus one
// cross-half exchange, and the rank-8 products (P V, dS K, P^T dO, dS^T Q) are vector FMAs against LDS rows
// that both tiles of a wave share.  Compared with the all-VALU kernels this removes 8 of 21 (forward),
// 16 of 30 (dQ) and 16 of 38 (dK/dV) vector instructions per (query, key) pair.
// Scores live in the log2 domain: log2(e)/sqrt(d) is folded into the Q (K) fragments, exp is v_exp_f32.
// Requirements: d in {8, 16}, L % 256 == 0 (others use attn.hip).  Deterministic, no atomics.
#include "common.h"
#include "h2_common.h"

namespace afd {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kTK = 64;                              // rows of the streamed operand per LDS tile
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, kWave); }     // value of lane ^ 32
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// rank-8 updates run as packed f32 FMAs (v_pk_fma_f32, the scalar broadcast through op_sel): measured 5.4 cycles per
// wave instruction against 4.3 for one v_fma_f32, i.e. 1.6x the FMA rate.  An LDS row of 8 is four register pairs.
using f2 = __attribute__((ext_vector_type(2))) float;
template <int D>
__device__ __forceinline__ void load_row(const float* __restrict__ p, f2 (&v)[D / 2]) {
#pragma unroll
  for (int q = 0; q < D / 4; ++q) {
    const float4 a = reinterpret_cast<const float4*>(p)[q];
    v[2 * q] = (f2){a.x, a.y}; v[2 * q + 1] = (f2){a.z, a.w};
  }
}
template <int D>
__device__ __forceinline__ void axpy(f2 (&acc)[D / 2], float s, const f2 (&v)[D / 2]) {
  const f2 ss = {s, s};
#pragma unroll
  for (int i = 0; i < D / 2; ++i) acc[i] = __builtin_elementwise_fma(ss, v[i], acc[i]);
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
// d = 16: a lane half carries head dims 8 half .. 8 half + 7, every fragment is one piece, six MFMAs per tile.
// B side (the row a lane keeps in registers for the whole kernel): d = 8: halves (0 | 1) carry  M1: b1 | b1,  M2: b2 | b1,
// M3: b2 | b3;  d = 16: f[m] = piece m of this half's eight dims.
template <int D>
__device__ __forceinline__ void row_frags(const float (&x)[D], int half, bf8 (&f)[3]) {
  bf8 p1, p2, p3;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 a, b, c;
    split3(D == 8 ? x[j] : (half ? x[(8 + j) % D] : x[j]), a, b, c);
    p1[j] = a; p2[j] = b; p3[j] = c;
  }
  if (D == 8) { f[0] = p1; f[1] = half ? p1 : p2; f[2] = half ? p3 : p2; }
  else { f[0] = p1; f[1] = p2; f[2] = p3; }
}
// A side (the streamed operand): pieces in LDS as 16-byte records (one ds_read_b128 per fragment) -- d = 8: [piece][row],
// halves carry  M1: a1 | a2,  M2: a1 | a3,  M3: a2 | a1;  d = 16: [piece][half][row].  A thread stages the d-pair
// (2 jp, 2 jp + 1) of row rr.
template <int D>
__device__ __forceinline__ void stage_pieces(uint32_t* __restrict__ P, int rr, int jp, float v0, float v1) {
  __bf16 a0, b0, c0, a1, b1, c1;
  split3(v0, a0, b0, c0); split3(v1, a1, b1, c1);
  const int rec = D == 8 ? rr : (jp >> 2) * kTK + rr, w = jp & 3, ps = (D / 8) * kTK;       // records per piece
  P[(0 * ps + rec) * 4 + w] = __builtin_bit_cast(uint32_t, (bf2){a0, a1});
  P[(1 * ps + rec) * 4 + w] = __builtin_bit_cast(uint32_t, (bf2){b0, b1});
  P[(2 * ps + rec) * 4 + w] = __builtin_bit_cast(uint32_t, (bf2){c0, c1});
}
template <int D>
__device__ __forceinline__ void load_frags(const uint32_t* __restrict__ P, int row, int half, bf8 (&a)[3]) {
  const bf8* p = reinterpret_cast<const bf8*>(P);
  if (D == 8) { a[0] = p[(half ? 1 : 0) * kTK + row]; a[1] = p[(half ? 2 : 0) * kTK + row]; a[2] = p[(half ? 0 : 1) * kTK + row]; }
  else {
#pragma unroll
    for (int m = 0; m < 3; ++m) a[m] = p[(m * 2 + half) * kTK + row];
  }
}
template <int D>
__device__ __forceinline__ f32x16 dotx3(const bf8 (&a)[3], const bf8 (&b)[3]) {
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  if (D == 8) {
#pragma unroll
    for (int m = 0; m < 3; ++m) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[m], c, 0, 0, 0);
  } else {                                                            // the six leading cross terms, one per MFMA
    constexpr int TA[6] = {0, 1, 0, 2, 1, 0}, TB[6] = {0, 0, 1, 0, 1, 2};
#pragma unroll
    for (int m = 0; m < 6; ++m) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[m]], b[TB[m]], c, 0, 0, 0);
  }
  return c;
}

// the same with the accumulator starting from c0 (a per-row constant folded into the product: S - lse, dP - delta)
template <int D>
__device__ __forceinline__ f32x16 dotx3c(const bf8 (&a)[3], const bf8 (&b)[3], f32x16 c) {
  static_assert(D == 8, "dotx3c: head dim 8");
#pragma unroll
  for (int m = 0; m < 3; ++m) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[m], c, 0, 0, 0);
  return c;
}

// ------------------------------------------------------------------------------------------------
// forward: workgroup = 256 queries (4 waves x 2 query tiles of 32), streams K/V in tiles of 64 keys
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void attn_fwd_mfma(const float* __restrict__ qkv, float* __restrict__ o,
                                                      float* __restrict__ lse, int heads, int L, float scale) {
  __shared__ __attribute__((aligned(16))) uint32_t Kp[3 * (D / 8) * kTK * 4];  // K pieces (MFMA A fragments)
  __shared__ __attribute__((aligned(16))) float Vr[kTK * D];       // V, row-major (P V rows)
  const int b = blockIdx.z, h = blockIdx.y, C = heads * D;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const int q0 = blockIdx.x * 256 + wv * 64;                          // this wave's first query
  bf8 bq[2][3];                                                       // B fragments of the two query tiles
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = qp[(long)d * L + q0 + j * 32 + l31] * (scale * kLog2e);
    row_frags<D>(x, half, bq[j]);
  }
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  f2 oa[2][D / 2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < D / 2; ++i) oa[j][i] = (f2){0.f, 0.f};

  // the next K / V tile travels global -> registers while the current one is multiplied: a thread owns the d-pairs
  // (2 jp, 2 jp + 1), jp = (tid >> 6) + 4 e, of row rr
  constexpr int NP = D / 8;
  const int jp0 = threadIdx.x >> 6, rr = threadIdx.x & 63;
  float kreg[NP][2], vreg[NP][2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < NP; ++e)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        kreg[e][h] = kp[(long)(2 * (jp0 + 4 * e) + h) * L + k0 + rr];
        vreg[e][h] = vp[(long)(2 * (jp0 + 4 * e) + h) * L + k0 + rr];
      }
  };
  fetch(0);
  for (int k0 = 0; k0 < L; k0 += kTK) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int jp = jp0 + 4 * e;
      stage_pieces<D>(Kp, rr, jp, kreg[e][0], kreg[e][1]);
      *reinterpret_cast<float2*>(Vr + rr * D + 2 * jp) = make_float2(vreg[e][0], vreg[e][1]);
    }
    __syncthreads();
    if (k0 + kTK < L) fetch(k0 + kTK);
#pragma unroll
    for (int kt = 0; kt < kTK / 32; ++kt) {
      bf8 ak[3];
      load_frags<D>(Kp, kt * 32 + l31, half, ak);
      f32x16 sc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        sc[j] = dotx3<D>(ak, bq[j]);
        float mx = sc[j][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[j][r]);
        mx = fmaxf(mx, xhalf(mx));
        const float mn = fmaxf(m[j], mx);
        const float alpha = __builtin_amdgcn_exp2f(m[j] - mn);        // m = -inf first: exp2(-inf) = 0
        l[j] *= alpha;
#pragma unroll
        for (int i = 0; i < D / 2; ++i) oa[j][i] *= alpha;
        m[j] = mn;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f2 v[D / 2];
        load_row<D>(Vr + (kt * 32 + acc_row(r, half)) * D, v);         // shared by both query tiles
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float p = __builtin_amdgcn_exp2f(sc[j][r] - m[j]);
          l[j] += p;
          axpy<D>(oa[j], p, v);
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
    for (int d = 0; d < D; ++d) {
      const float mine = oa[j][d >> 1][d & 1];
      const float t = mine + xhalf(mine);
      if (half == 0) o[((long)b * C + h * D + d) * L + qi] = t * inv;
    }
    if (half == 0) lse[((long)b * heads + h) * L + qi] = (m[j] + __builtin_amdgcn_logf(lt)) * kLn2;   // v_log_f32 = log2
  }
}

// P' = exp2(s - off) for the 16 scores of a lane, split into two fp16 pieces as the B operands of two (k = registers 0..7,
// 8..15) x two (piece) MFMAs: one packed conversion per pair and piece
using h2v = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
// -1.0 in a scalar register the optimizer cannot see through: fma(piece, -1, x) with a literal -1 is folded back into a
// conversion + subtraction; with an opaque factor it stays one v_fma_mix_f32 (the fp16 piece is read in place)
__device__ __forceinline__ float opaque_neg1() {
  float v;
  asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(v));
  return v;
}
__device__ __forceinline__ void pv_split16(const f32x16& sc, float off, float neg1, h8 (&p1)[2], h8 (&p2)[2]) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    u32x4 w1, w2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float x0 = __builtin_amdgcn_exp2f(sc[8 * g + 2 * q] - off), x1 = __builtin_amdgcn_exp2f(sc[8 * g + 2 * q + 1] - off);
      const h2v a = __builtin_convertvector((f2){x0, x1}, h2v);
      const h2v c = __builtin_convertvector((f2){__builtin_fmaf((float)a[0], neg1, x0), __builtin_fmaf((float)a[1], neg1, x1)}, h2v);
      w1[q] = __builtin_bit_cast(uint32_t, a);
      w2[q] = __builtin_bit_cast(uint32_t, c);
    }
    p1[g] = __builtin_bit_cast(h8, w1);
    p2[g] = __builtin_bit_cast(h8, w2);
  }
}

// ------------------------------------------------------------------------------------------------
// forward, round 3: P V on the fp16 matrix pipe too.
// The kernel above is bound by vector instructions: per (query, key) pair a max, a sub, a quarter-rate exp, an add (the
// row sum) and four packed FMAs (the rank-8 update O += p V[key]) = 12 issue slots.  Here the S^T tile that the first MFMA
// leaves in the accumulator layout (lane = query, 16 registers = 16 keys) is ITSELF the B operand of a second product:
//   O'^T[m][query] += sum_key A[m][key] * P'[key][query],   v_mfma_f32_32x32x16_f16, k-slot (half, j) <-> key acc_row(8 g + j, half)
// with P' = 2^14 P split into two fp16 pieces (P <= 1: 22 bits; the factor cancels against the row sum) and the rows of A
//   m = 0..7: V piece 1 (d = m),  m = 8..15: V piece 2,  m = 16, 20: ones (their output row IS the row sum),  others 0.
// V is scaled by a power of two that follows the data (h2_common.h: the per-stage maximum; when it would overflow fp16 the
// scale is lowered and the running sums carried over).  Four MFMAs per 32 x 32 tile replace 64 packed FMAs + 16 adds; what is
// left on the vector pipe per pair is max/2 + sub + exp + ~2 conversion slots: 7.5 instead of 12.  The piece products
// v1 p1 + v2 p1 + v1 p2 (+ v2 p2) are exact in the fp32 accumulator; the result is fp32-class (2^-22 per product).
// Output: lane (query, half) ends with d = 4 half .. 4 half + 3 of its query -- no cross-lane step.
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void attn_fwd_pv(const float* __restrict__ qkv, float* __restrict__ o,
                                                    float* __restrict__ lse, int heads, int L, float scale) {
  static_assert(D == 8, "attn_fwd_pv: head dim 8");
  __shared__ __attribute__((aligned(16))) uint32_t Kp[3 * kTK * 4];            // K pieces (A fragments of S^T)
  __shared__ __attribute__((aligned(16))) _Float16 Va[2 * 2 * 2 * 32 * 8];     // [key tile][g][half][m][8]: A fragments of P V
  __shared__ float wmax[4];
  const int b = blockIdx.z, h = blockIdx.y, C = heads * D;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const int q0 = blockIdx.x * 256 + wv * 64;
  bf8 bq[2][3];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = qp[(long)d * L + q0 + j * 32 + l31] * (scale * kLog2e);
    row_frags<D>(x, half, bq[j]);
  }
  // the constant rows of A: ones at m = 16 and 20 (the row sum lands in accumulator register 8 of both lane halves), else 0
  for (int i = threadIdx.x; i < 2 * 2 * 2 * 16 * 8; i += 256) {
    const int rec = i >> 3, m = 16 + (rec & 15), blk = rec >> 4;
    Va[(blk * 32 + m) * 8 + (i & 7)] = (m == 16 || m == 20) ? (_Float16)1.0f : (_Float16)0.0f;
  }
  float m[2] = {-INFINITY, -INFINITY};
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float sv = __uint_as_float(kH2ScaleCapBits);                        // running power-of-two scale of V
  const float neg1 = opaque_neg1();

  const int jp0 = threadIdx.x >> 6, rr = threadIdx.x & 63;           // staging: d-pair (2 jp0, 2 jp0 + 1) of key row rr
  // where this thread's V elements go: key rr = 32 kt + rho, rho = (r & 3) + 8 (r >> 2) + 4 half' -> k-slot (half', j = r & 7) of MFMA g = r >> 3
  const int rho = rr & 31, vh = (rho >> 2) & 1, vr = (rho & 3) + 4 * (rho >> 3);
  const int vbase = ((((rr >> 5) * 2 + (vr >> 3)) * 2 + vh) * 32) * 8 + (vr & 7);       // + m * 8
  float kreg[2], vreg[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      kreg[hh] = kp[(long)(2 * jp0 + hh) * L + k0 + rr];
      vreg[hh] = vp[(long)(2 * jp0 + hh) * L + k0 + rr];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < L; k0 += kTK) {
    {
      const float mv = wave_amax(fmaxf(fabsf(vreg[0]), fabsf(vreg[1])));
      if (lane == 0) wmax[wv] = mv;
    }
    __syncthreads();                                                   // the previous stage's fragment reads are done; the maxima are visible
    {
      const float sn = fminf(sv, h2_scale_for(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
      if (sn != sv) {                                                  // (uniform) carry the V-weighted sums over to the lower scale
        const float f = sn * h2_inv_pow2(sv);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 8; ++r) acc[j][r] *= f;
        sv = sn;
      }
    }
    stage_pieces<D>(Kp, rr, jp0, kreg[0], kreg[1]);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      _Float16 a, c;
      h2_split(vreg[hh], sv, a, c);
      Va[vbase + (2 * jp0 + hh) * 8] = a;
      Va[vbase + (2 * jp0 + hh + 8) * 8] = c;
    }
    __syncthreads();
    if (k0 + kTK < L) fetch(k0 + kTK);
#pragma unroll
    for (int kt = 0; kt < kTK / 32; ++kt) {
      bf8 ak[3];
      load_frags<D>(Kp, kt * 32 + l31, half, ak);
      const h8* va = reinterpret_cast<const h8*>(Va) + ((kt * 2) * 2 + half) * 32 + l31;
      const h8 av0 = va[0], av1 = va[2 * 32];                          // g = 0, 1
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 sc = dotx3<D>(ak, bq[j]);
        float mx = fmaxf(sc[0], sc[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sc[r]), sc[r + 1]);
        mx = fmaxf(mx, xhalf(mx));
        const float mn = fmaxf(m[j], mx);
        const float alpha = __builtin_amdgcn_exp2f(m[j] - mn);         // m = -inf first: exp2(-inf) = 0
        m[j] = mn;
#pragma unroll
        for (int r = 0; r < 9; ++r) acc[j][r] *= alpha;                // eight sums + the row sum
        const float off = mn - 14.0f;                                  // P' = 2^14 P
        h8 p1[2], p2[2];
        pv_split16(sc, off, neg1, p1, p2);
        __builtin_amdgcn_s_setprio(2);                                 // matrix-pipe phases issue ahead of the other wave's vector work
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av0, p1[0], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av1, p1[1], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av0, p2[0], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av1, p2[1], acc[j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }
  }
  const float isv = h2_inv_pow2(sv);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float lt = acc[j][8];                                        // 2^14 x the softmax denominator
    const float inv = isv / lt;
    const int qi = q0 + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      o[((long)b * C + h * D + 4 * half + i) * L + qi] = (acc[j][i] + acc[j][4 + i]) * inv;
    if (half == 0) lse[((long)b * heads + h) * L + qi] = (m[j] - 14.0f + __builtin_amdgcn_logf(lt)) * kLn2;   // v_log_f32 = log2
  }
}

// ------------------------------------------------------------------------------------------------
// dQ (and delta = rowsum(dO * O)): tiling of the forward with TJ query tiles per wave (d = 16: one -- 246 VGPRs at d = 8
// leave no room for 16-wide rows of a second tile); S^T and dP^T on MFMA
// ------------------------------------------------------------------------------------------------
template <int D, int TJ>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_mfma(const float* __restrict__ qkv, const float* __restrict__ o,
                                                         const float* __restrict__ d_o, const float* __restrict__ lse,
                                                         float* __restrict__ dqkv, float* __restrict__ delta_out,
                                                         int heads, int L, float scale) {
  __shared__ __attribute__((aligned(16))) uint32_t Kp[3 * (D / 8) * kTK * 4];  // K pieces (A fragments of S^T)
  __shared__ __attribute__((aligned(16))) uint32_t Vp[3 * (D / 8) * kTK * 4];  // V pieces (A fragments of dP^T)
  __shared__ __attribute__((aligned(16))) float Kr[kTK * D];
  const int b = blockIdx.z, h = blockIdx.y, C = heads * D;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const long ob = ((long)b * C + h * D) * L;
  const int q0 = blockIdx.x * (128 * TJ) + wv * (32 * TJ);
  bf8 bq[TJ][3], bg[TJ][3];
  float lsq[TJ], dlt[TJ];
  f2 dq[TJ][D / 2];
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int qi = q0 + j * 32 + l31;
    float dpart = 0.f, xq[D], xg[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      xq[d] = qp[(long)d * L + qi] * (scale * kLog2e);
      xg[d] = d_o[ob + (long)d * L + qi];
      dpart = fmaf(xg[d], o[ob + (long)d * L + qi], dpart);
    }
    row_frags<D>(xq, half, bq[j]);
    row_frags<D>(xg, half, bg[j]);
    dlt[j] = dpart;
    lsq[j] = lse[((long)b * heads + h) * L + qi] * kLog2e;
    if (half == 0) delta_out[((long)b * heads + h) * L + qi] = dlt[j];
#pragma unroll
    for (int i = 0; i < D / 2; ++i) dq[j][i] = (f2){0.f, 0.f};
  }
  constexpr int NP = D / 8;                             // staging: d-pairs (2 jp, 2 jp + 1), jp = (tid >> 6) + 4 e, of row rr
  const int jp0 = threadIdx.x >> 6, rr = threadIdx.x & 63;
  float kreg[NP][2], vreg[NP][2];                       // next K / V tile in flight during the multiplies
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < NP; ++e)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        kreg[e][h] = kp[(long)(2 * (jp0 + 4 * e) + h) * L + k0 + rr];
        vreg[e][h] = vp[(long)(2 * (jp0 + 4 * e) + h) * L + k0 + rr];
      }
  };
  fetch(0);
  for (int k0 = 0; k0 < L; k0 += kTK) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int jp = jp0 + 4 * e;
      stage_pieces<D>(Kp, rr, jp, kreg[e][0], kreg[e][1]);
      stage_pieces<D>(Vp, rr, jp, vreg[e][0], vreg[e][1]);
      *reinterpret_cast<float2*>(Kr + rr * D + 2 * jp) = make_float2(kreg[e][0], kreg[e][1]);
    }
    __syncthreads();
    if (k0 + kTK < L) fetch(k0 + kTK);
#pragma unroll
    for (int kt = 0; kt < kTK / 32; ++kt) {
      bf8 ak[3], av[3];
      load_frags<D>(Kp, kt * 32 + l31, half, ak);
      load_frags<D>(Vp, kt * 32 + l31, half, av);
      f32x16 sc[TJ], dp[TJ];
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        sc[j] = dotx3<D>(ak, bq[j]);
        dp[j] = dotx3<D>(av, bg[j]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f2 kr[D / 2];
        load_row<D>(Kr + (kt * 32 + acc_row(r, half)) * D, kr);
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const float ds = __builtin_amdgcn_exp2f(sc[j][r] - lsq[j]) * (dp[j][r] - dlt[j]);
          axpy<D>(dq[j], ds, kr);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int qi = q0 + j * 32 + l31;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float mine = dq[j][d >> 1][d & 1];
      const float t = mine + xhalf(mine);
      if (half == 0) dqkv[((long)b * 3 * C + h * D + d) * L + qi] = t * scale;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: workgroup = 4 waves x TJ key tiles of 32 (d = 8: TJ = 2, 256 keys; d = 16: TJ = 1, 128 keys -- the resident
// fragments, the dK / dV sums and the score tiles of a second tile do not fit beside 16-wide rows), streams Q / dO / lse /
// delta in tiles of 64 queries.  Tiles are [query rows x key columns]: lane = key, registers = 16 of 32 queries.
// ------------------------------------------------------------------------------------------------
template <int D, int TJ>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_mfma(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ dqkv, int heads, int L, float scale) {
  __shared__ __attribute__((aligned(16))) uint32_t Qp[3 * (D / 8) * kTK * 4];  // Q pieces  (A fragments of S)
  __shared__ __attribute__((aligned(16))) uint32_t Gp[3 * (D / 8) * kTK * 4];  // dO pieces (A fragments of dP)
  __shared__ __attribute__((aligned(16))) float Qr[kTK * D];       // rows for dK += dS^T Q
  __shared__ __attribute__((aligned(16))) float Gr[kTK * D];       // rows for dV += P^T dO
  __shared__ float Ls[kTK];
  __shared__ float Ds[kTK];
  const int b = blockIdx.z, h = blockIdx.y, C = heads * D;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const float* gp = d_o + ((long)b * C + h * D) * L;
  const float* lp = lse + ((long)b * heads + h) * L;
  const float* dlp = delta + ((long)b * heads + h) * L;
  const int key0 = blockIdx.x * (128 * TJ) + wv * (32 * TJ);
  // B fragments of the wave's key tiles (K scaled, V): a wave-private LDS image, one ds_read_b128 per fragment and
  // lane (in registers they cost 48 VGPRs, and the kernel must keep 2 waves / SIMD)
  __shared__ bf8 Bf[4][TJ][2][3][64];
  f2 dk[TJ][D / 2], dv[TJ][D / 2];
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    float xk[D], xv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      xk[d] = kp[(long)d * L + key0 + j * 32 + l31] * (scale * kLog2e);
      xv[d] = vp[(long)d * L + key0 + j * 32 + l31];
    }
    bf8 t[3];
    row_frags<D>(xk, half, t);
#pragma unroll
    for (int m = 0; m < 3; ++m) Bf[wv][j][0][m][lane] = t[m];
    row_frags<D>(xv, half, t);
#pragma unroll
    for (int m = 0; m < 3; ++m) Bf[wv][j][1][m][lane] = t[m];
#pragma unroll
    for (int i = 0; i < D / 2; ++i) { dk[j][i] = (f2){0.f, 0.f}; dv[j][i] = (f2){0.f, 0.f}; }
  }
  constexpr int NP = D / 8;                                 // staging: d-pairs (2 jp, 2 jp + 1), jp = (tid >> 6) + 4 e, of row rr
  const int jp0 = threadIdx.x >> 6, rr = threadIdx.x & 63;
  float qreg[NP][2], greg[NP][2], lreg = 0.f, dreg = 0.f;   // next Q / dO / lse / delta tile in flight during the multiplies
  auto fetch = [&](int t0) {
#pragma unroll
    for (int e = 0; e < NP; ++e)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        qreg[e][hh] = qp[(long)(2 * (jp0 + 4 * e) + hh) * L + t0 + rr];
        greg[e][hh] = gp[(long)(2 * (jp0 + 4 * e) + hh) * L + t0 + rr];
      }
    if (threadIdx.x < kTK) { lreg = lp[t0 + threadIdx.x] * kLog2e; dreg = dlp[t0 + threadIdx.x]; }
  };
  fetch(0);
  for (int t0 = 0; t0 < L; t0 += kTK) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int jp = jp0 + 4 * e;
      stage_pieces<D>(Qp, rr, jp, qreg[e][0], qreg[e][1]);
      stage_pieces<D>(Gp, rr, jp, greg[e][0], greg[e][1]);
      *reinterpret_cast<float2*>(Qr + rr * D + 2 * jp) = make_float2(qreg[e][0], qreg[e][1]);
      *reinterpret_cast<float2*>(Gr + rr * D + 2 * jp) = make_float2(greg[e][0], greg[e][1]);
    }
    if (threadIdx.x < kTK) { Ls[threadIdx.x] = lreg; Ds[threadIdx.x] = dreg; }
    __syncthreads();
    if (t0 + kTK < L) fetch(t0 + kTK);
#pragma unroll
    for (int qt = 0; qt < kTK / 32; ++qt) {
      bf8 aq[3], ag[3];
      load_frags<D>(Qp, qt * 32 + l31, half, aq);
      load_frags<D>(Gp, qt * 32 + l31, half, ag);
      f32x16 sc[TJ], dp[TJ];
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        bf8 bkj[3], bvj[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) { bkj[m] = Bf[wv][j][0][m][lane]; bvj[m] = Bf[wv][j][1][m][lane]; }
        sc[j] = dotx3<D>(aq, bkj);
        dp[j] = dotx3<D>(ag, bvj);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = qt * 32 + acc_row(r, half);
        f2 qrow[D / 2], grow[D / 2];
        load_row<D>(Qr + qr * D, qrow);
        load_row<D>(Gr + qr * D, grow);
        const float ls = Ls[qr], dl = Ds[qr];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const float p = __builtin_amdgcn_exp2f(sc[j][r] - ls);
          const float ds = p * (dp[j][r] - dl);
          axpy<D>(dv[j], p, grow);
          axpy<D>(dk[j], ds, qrow);
        }
        if ((r & 1) == 1) asm volatile("" ::: "memory");      // bound how many LDS rows are in flight (register budget)
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int ki = key0 + j * 32 + l31;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float mk = dk[j][d >> 1][d & 1], mv = dv[j][d >> 1][d & 1];
      const float tk = mk + xhalf(mk), tv = mv + xhalf(mv);
      if (half == 0) {
        dqkv[((long)b * 3 * C + C + h * D + d) * L + ki] = tk * scale;
        dqkv[((long)b * 3 * C + 2 * C + h * D + d) * L + ki] = tv;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward in ONE pass, round 3 (d = 8): every product on the matrix pipes.
//
// Round 2's two passes each recomputed S, dP and the exponentials and ran their rank-8 updates as packed vector FMAs: 12 + 17
// vector issue slots per (query, key) pair.  Here a workgroup owns 256 keys of one (batch, head) at a time (a wave: two tiles
// of 32), streams the queries in stages of 64 and computes S / dP (bf16x3 MFMAs, as before), P' = 2^14 P and dS' = s_ds dS
// ONCE, in the layout lane = key, 16 registers = 16 queries.  Both tiles, split into two fp16 pieces, are then operands:
//   dV'^T[m][key] += sum_q GA[m][q] P'[q][key]      A rows: dO pieces (m = d, d + 8)         } the tile is the B operand:
//   dK'^T[m][key] += sum_q QA[m][q] dS'[q][key]     A rows: Q pieces                          } contraction over registers
//   dS'^T = sum_g dS'_g as A x (a 0/1 selection matrix as B): the SAME registers read as an A operand give the tile with the
//           roles of lanes and registers swapped -- the matrix pipe transposes it, exactly (one non-zero product per output);
//   dQ'^T[m][query] += sum_key KA[m][key] dS'^T[key][query]                                    (its pieces re-packed: 16 cvt)
// 22 MFMAs (32x32x16) per 32 x 32 tile against ~15 vector slots per pair (exp, the softmax algebra, three fp16 splits): half the
// vector work of the two passes.  Scales (powers of two, h2_common.h): Q, dO per workgroup (running, from each stage's
// maximum), K per key block and wave, dS per wave from the bound |dS| <= max_key |V|_1 max|dO| + max|delta|; every change
// rescales the affected accumulators (exact).  dQ: each wave's partial for a 32-query tile (its 64 keys) is unscaled, the four
// waves' partials meet in LDS in wave order and are added to global memory by this workgroup alone, key block after key block
// (no other workgroup touches this (batch, head)): deterministic, no atomics, no slabs.  delta = rowsum(dO o O) comes from a
// small kernel ahead.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_k(const float* __restrict__ o, const float* __restrict__ d_o, float* __restrict__ delta,
                                                    int heads, int D, int L, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;                 // (b, h, q)
  if (i >= total) return;
  const int q = i % L; const long bh = i / L;
  const float* po = o + bh * D * L + q;
  const float* pg = d_o + bh * D * L + q;
  float s = 0.f;
  for (int d = 0; d < D; ++d) s = fmaf(pg[(long)d * L], po[(long)d * L], s);
  delta[i] = s;
}

// two fp16 pieces of 16 fp32 values (the registers of one accumulator tile) as the operands of the two k-groups
__device__ __forceinline__ void split16(const f32x16& v, float neg1, h8 (&p1)[2], h8 (&p2)[2]) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    u32x4 w1, w2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float x0 = v[8 * g + 2 * q], x1 = v[8 * g + 2 * q + 1];
      const h2v a = __builtin_convertvector((f2){x0, x1}, h2v);
      const h2v c = __builtin_convertvector((f2){__builtin_fmaf((float)a[0], neg1, x0), __builtin_fmaf((float)a[1], neg1, x1)}, h2v);   // (v_fma_mix_f32, opaque_neg1 above)
      w1[q] = __builtin_bit_cast(uint32_t, a);
      w2[q] = __builtin_bit_cast(uint32_t, c);
    }
    p1[g] = __builtin_bit_cast(h8, w1);
    p2[g] = __builtin_bit_cast(h8, w2);
  }
}
__device__ __forceinline__ void pack16(const f32x16& v, h8 (&p)[2]) {     // fp32 registers that hold fp16 values exactly
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    u32x4 w;
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){v[8 * g + 2 * q], v[8 * g + 2 * q + 1]}, h2v));
    p[g] = __builtin_bit_cast(h8, w);
  }
}

template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_fused(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                       const float* __restrict__ lse, const float* __restrict__ delta,
                                                       float* __restrict__ dqkv, int heads, int L, float scale) {
  static_assert(D == 8, "attn_bwd_fused: head dim 8");
  constexpr int REC = 2 * 2 * 2 * 16;                                  // A-operand records (query / key tile, g, half, m < 16) of one set
  __shared__ __attribute__((aligned(16))) uint32_t Qp[3 * kTK * 4];   // Q pieces  (bf16x3 A fragments of S)
  __shared__ __attribute__((aligned(16))) uint32_t Gp[3 * kTK * 4];   // dO pieces (bf16x3 A fragments of dP)
  __shared__ __attribute__((aligned(16))) _Float16 QA[(REC + 1) * 8]; // fp16 A operands of dK: rows m = d (piece 1), d + 8 (piece 2); + one zero record
  __shared__ __attribute__((aligned(16))) _Float16 GA[(REC + 1) * 8]; // ... of dV (dO pieces)
  __shared__ __attribute__((aligned(16))) _Float16 KA[4 * (REC + 1) * 8];   // ... of dQ: this wave's K pieces (per wave)
  __shared__ __attribute__((aligned(16))) bf8 Bf[4][2][2][3][64];     // bf16x3 B fragments of the wave's key tiles: K (scaled), V
  __shared__ __attribute__((aligned(16))) float Ls[kTK];
  __shared__ __attribute__((aligned(16))) float Ds[kTK];
  __shared__ float DQ[4][D][kTK];                                      // the waves' dQ partials of one stage
  __shared__ float wmax[4][4];
  const int b = blockIdx.y, h = blockIdx.x, C = heads * D;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const float* gp = d_o + ((long)b * C + h * D) * L;
  const float* lp = lse + ((long)b * heads + h) * L;
  const float* dlp = delta + ((long)b * heads + h) * L;
  float* dqg = dqkv + ((long)b * 3 * C + h * D) * L;

  // zero records (rows m >= 16 of every A operand) and the selection matrices of the transposition:
  // Perm[g]: k-slot (half, j) selects column n = acc_row(8 g + j, half)
  if (threadIdx.x < 8) { QA[REC * 8 + threadIdx.x] = (_Float16)0.f; GA[REC * 8 + threadIdx.x] = (_Float16)0.f; }
  if (threadIdx.x < 32) KA[(threadIdx.x >> 3) * (REC + 1) * 8 + REC * 8 + (threadIdx.x & 7)] = (_Float16)0.f;
  h8 perm[2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) perm[g][j] = (l31 == acc_row(8 * g + j, half)) ? (_Float16)1.0f : (_Float16)0.0f;
  // A-operand record of this lane for (tile t, group g): rows m >= 16 read the zero record
  auto a_rec = [&](int t, int g) { return l31 < 16 ? ((t * 2 + g) * 2 + half) * 16 + l31 : REC; };
  // where a staged row's elements go: row rr = 32 t + rho, rho = (r & 3) + 8 (r >> 2) + 4 half' -> k-slot (half', r & 7) of group r >> 3
  const int jp0 = threadIdx.x >> 6, rr = threadIdx.x & 63;
  const int rho = rr & 31, sh = (rho >> 2) & 1, sr = (rho & 3) + 4 * (rho >> 3);
  const int sbase = ((((rr >> 5) * 2 + (sr >> 3)) * 2 + sh) * 16) * 8 + (sr & 7);         // + m * 8, m = d or d + 8

  float sq = __uint_as_float(kH2ScaleCapBits), sg = sq;                // running scales of Q and dO (workgroup)
  const float neg1 = opaque_neg1();
  for (int kb = 0; kb < L; kb += 256) {
    const int key0 = kb + wv * 64;
    // ---- this wave's keys: bf16x3 B fragments (K scaled for the log2-domain scores, V), fp16 A operand of dQ (raw K), |V|_1
    float v1 = 0.f, kmax = 0.f;
    float xk[2][D];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float xs[D], xv[D];
      float n1 = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        xk[j][d] = kp[(long)d * L + key0 + j * 32 + l31];
        xs[d] = xk[j][d] * (scale * kLog2e);
        xv[d] = vp[(long)d * L + key0 + j * 32 + l31];
        n1 += fabsf(xv[d]); kmax = fmaxf(kmax, fabsf(xk[j][d]));
      }
      v1 = fmaxf(v1, n1);
      bf8 t[3];
      row_frags<D>(xs, half, t);
#pragma unroll
      for (int m = 0; m < 3; ++m) Bf[wv][j][0][m][lane] = t[m];
      row_frags<D>(xv, half, t);
#pragma unroll
      for (int m = 0; m < 3; ++m) Bf[wv][j][1][m][lane] = t[m];
    }
    v1 = wave_amax(v1);
    const float sk = h2_scale_for(wave_amax(kmax));                    // this wave's K scale for the block
    if (half == 0) {                                                   // key l31 of tile j: k-slot as above
      const int kh = (l31 >> 2) & 1, kr = (l31 & 3) + 4 * (l31 >> 3);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int d = 0; d < D; ++d) {
          _Float16 a, c;
          h2_split(xk[j][d], sk, a, c);
          const int base = wv * (REC + 1) * 8 + (((j * 2 + (kr >> 3)) * 2 + kh) * 16) * 8 + (kr & 7);
          KA[base + d * 8] = a; KA[base + (d + 8) * 8] = c;
        }
    }
    f32x16 accV[2], accK[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accV[j][r] = 0.f; accK[j][r] = 0.f; }
    float sds = __uint_as_float(kH2ScaleCapBits);                      // the running scale of dS for this key block (workgroup-wide: it rides in the staged dO pieces)

    float qreg[2], greg[2], lreg = 0.f, dreg = 0.f;
    auto fetch = [&](int t0) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        qreg[hh] = qp[(long)(2 * jp0 + hh) * L + t0 + rr];
        greg[hh] = gp[(long)(2 * jp0 + hh) * L + t0 + rr];
      }
      if (threadIdx.x < kTK) { lreg = lp[t0 + threadIdx.x] * kLog2e; dreg = dlp[t0 + threadIdx.x]; }
    };
    // the four waves' dQ partials of the stage that ended: added in wave order, into global memory (first key block: stored)
    auto dq_flush = [&](int t0) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int d = threadIdx.x >> 5, q = (threadIdx.x & 31) + 32 * it;
        const float t = ((DQ[0][d][q] + DQ[1][d][q]) + DQ[2][d][q]) + DQ[3][d][q];
        float* dst = dqg + (long)d * L + t0 + q;
        *dst = kb == 0 ? t : *dst + t;
      }
    };
    fetch(0);
    for (int t0 = 0; t0 < L; t0 += kTK) {
      {
        const float mq = wave_amax(fmaxf(fabsf(qreg[0]), fabsf(qreg[1])));
        const float mg = wave_amax(fmaxf(fabsf(greg[0]), fabsf(greg[1])));
        const float md = wave_amax(fabsf(dreg));                       // (threads >= 64 hold 0)
        if (lane == 0) { wmax[0][wv] = mq; wmax[1][wv] = mg; wmax[2][wv] = md; wmax[3][wv] = v1; }
      }
      __syncthreads();                                                 // A: the previous stage's reads are done, its dQ partials and the maxima visible
      const float mq = fmaxf(fmaxf(wmax[0][0], wmax[0][1]), fmaxf(wmax[0][2], wmax[0][3]));
      const float mg = fmaxf(fmaxf(wmax[1][0], wmax[1][1]), fmaxf(wmax[1][2], wmax[1][3]));
      const float md = fmaxf(fmaxf(wmax[2][0], wmax[2][1]), fmaxf(wmax[2][2], wmax[2][3]));
      const float v1g = fmaxf(fmaxf(wmax[3][0], wmax[3][1]), fmaxf(wmax[3][2], wmax[3][3]));   // max_key |V|_1 over the block's 256 keys
      {
        const float sqn = fminf(sq, h2_scale_for(mq)), sgn = fminf(sg, h2_scale_for(mg));
        const float sdn = fminf(sds, h2_scale_for(fmaf(v1g, mg, md))); // |dS| <= |V|_1 max|dO| + max|delta|
        // (and c max|dO| stays finite where V and delta vanish: blocks whose dS is below 2^-49 max|dO| are not resolved further)
        const float sdc = fminf(sdn, h2_scale_for(mg) * 0x1p64f);
        const float fk = (sqn * h2_inv_pow2(sq)) * (sdc * h2_inv_pow2(sds)), fv = sgn * h2_inv_pow2(sg);
        if (fk != 1.0f || fv != 1.0f) {                                // (uniform per wave)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 8; ++r) { accK[j][r] *= fk; accV[j][r] *= fv; }
        }
        sq = sqn; sg = sgn; sds = sdc;
      }
      // dS' = P' (dP - delta) c, c = s_ds 2^-14 (P' = 2^14 P): the factor c rides in the staged dO pieces (a power of two: exact)
      // and both per-query constants start the accumulators of their products -- S - (lse log2(e) - 14) and c dP - c delta leave
      // the matrix pipe ready, one exp2 and one multiply per pair remain on the vector pipe
      const float cds = sds * (1.0f / 16384.0f);
      stage_pieces<D>(Qp, rr, jp0, qreg[0], qreg[1]);
      stage_pieces<D>(Gp, rr, jp0, greg[0] * cds, greg[1] * cds);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        _Float16 a, c;
        h2_split(qreg[hh], sq, a, c);
        QA[sbase + (2 * jp0 + hh) * 8] = a; QA[sbase + (2 * jp0 + hh + 8) * 8] = c;
        h2_split(greg[hh], sg, a, c);
        GA[sbase + (2 * jp0 + hh) * 8] = a; GA[sbase + (2 * jp0 + hh + 8) * 8] = c;
      }
      if (threadIdx.x < kTK) Ls[threadIdx.x] = 14.0f - lreg;           // (negated: the rows are accumulator start values)
      if (t0 > 0) dq_flush(t0 - kTK);
      if (threadIdx.x < kTK) Ds[threadIdx.x] = -dreg * cds;
      __syncthreads();                                                 // B
      if (t0 + kTK < L) fetch(t0 + kTK);
      const float udq = (scale * h2_inv_pow2(sds)) * h2_inv_pow2(sk);  // dQ partials leave unscaled
#pragma unroll
      for (int qt = 0; qt < kTK / 32; ++qt) {
        const h8* qa = reinterpret_cast<const h8*>(QA);
        const h8* ga = reinterpret_cast<const h8*>(GA);
        const h8* ka = reinterpret_cast<const h8*>(KA) + wv * (REC + 1);
        f32x16 accQ;
#pragma unroll
        for (int r = 0; r < 16; ++r) accQ[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // (operands are fetched from LDS right where they are used: the accumulators of five products leave few registers)
          f32x16 sc, dp;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {                             // registers 4 r4 .. 4 r4 + 3 are four consecutive query rows
            const int qr = qt * 32 + 8 * r4 + 4 * half;
            const float4 ls = *reinterpret_cast<const float4*>(Ls + qr), dl = *reinterpret_cast<const float4*>(Ds + qr);
            sc[4 * r4] = ls.x; sc[4 * r4 + 1] = ls.y; sc[4 * r4 + 2] = ls.z; sc[4 * r4 + 3] = ls.w;
            dp[4 * r4] = dl.x; dp[4 * r4 + 1] = dl.y; dp[4 * r4 + 2] = dl.z; dp[4 * r4 + 3] = dl.w;
          }
          {
            bf8 aq[3], bkj[3];
            load_frags<D>(Qp, qt * 32 + l31, half, aq);
#pragma unroll
            for (int m = 0; m < 3; ++m) bkj[m] = Bf[wv][j][0][m][lane];
            sc = dotx3c<D>(aq, bkj, sc);
          }
          {
            bf8 ag[3], bvj[3];
            load_frags<D>(Gp, qt * 32 + l31, half, ag);
#pragma unroll
            for (int m = 0; m < 3; ++m) bvj[m] = Bf[wv][j][1][m][lane];
            dp = dotx3c<D>(ag, bvj, dp);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(sc[r]);                               // P' = 2^14 P
            sc[r] = pv;
            dp[r] *= pv;                                                                  // dS'
          }
          __builtin_amdgcn_sched_barrier(0);
          h8 p1[2], p2[2];
          {
            split16(sc, neg1, p1, p2);
            const h8 ga0 = ga[a_rec(qt, 0)], ga1 = ga[a_rec(qt, 1)];
            __builtin_amdgcn_s_setprio(2);                             // matrix-pipe phases issue ahead of the other wave's vector work
            accV[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga0, p1[0], accV[j], 0, 0, 0);
            accV[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga1, p1[1], accV[j], 0, 0, 0);
            accV[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga0, p2[0], accV[j], 0, 0, 0);
            accV[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga1, p2[1], accV[j], 0, 0, 0);
          }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          {
            split16(dp, neg1, p1, p2);
            const h8 qa0 = qa[a_rec(qt, 0)], qa1 = qa[a_rec(qt, 1)];
            __builtin_amdgcn_s_setprio(2);
            accK[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa0, p1[0], accK[j], 0, 0, 0);
            accK[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa1, p1[1], accK[j], 0, 0, 0);
            accK[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa0, p2[0], accK[j], 0, 0, 0);
            accK[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa1, p2[1], accK[j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          // the dS' pieces again, now with lane = query, registers = keys: the tile read as an A operand times the selection
          // matrices (exact: one non-zero product per output), one piece at a time
          const h8 ka0 = ka[a_rec(j, 0)], ka1 = ka[a_rec(j, 1)];
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) {
            f32x16 t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = 0.f;
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(pc ? p2[0] : p1[0], perm[0], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(pc ? p2[1] : p1[1], perm[1], t, 0, 0, 0);
            h8 e[2];
            pack16(t, e);
            accQ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka0, e[0], accQ, 0, 0, 0);
            accQ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka1, e[1], accQ, 0, 0, 0);
          }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) DQ[wv][4 * half + i][qt * 32 + l31] = (accQ[i] + accQ[4 + i]) * udq;
      }
    }
    __syncthreads();                                                   // the last stage's dQ partials
    dq_flush(L - kTK);
    // ---- dK, dV of this key block: lane (key, half) holds d = 4 half .. 4 half + 3
    {
      const float uk = (scale * h2_inv_pow2(sds)) * h2_inv_pow2(sq), uv = h2_inv_pow2(sg) * (1.0f / 16384.0f);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ki = key0 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dqkv[((long)b * 3 * C + C + h * D + 4 * half + i) * L + ki] = (accK[j][i] + accK[j][4 + i]) * uk;
          dqkv[((long)b * 3 * C + 2 * C + h * D + 4 * half + i) * L + ki] = (accV[j][i] + accV[j][4 + i]) * uv;
        }
      }
    }
    __syncthreads();                                                   // DQ and the wave-resident images are free for the next key block
  }
}

// A one-pass backward (S, dP, the exponentials and dS computed once, P and dS transposed through wave-private LDS so that
// dK / dV become in-lane sums) was built and measured in round 1: register-bound (the loop-invariant Q / dO rows take
// ~250 registers, or spill), 2.7-13x slower than the two passes above.  It was removed in round 2; the numbers are in
// DESIGN.md section 6.

// host-side launchers used by attn.hip
bool attn_mfma8_ok(int d, int L) { return (d == 8 || d == 16) && L % 256 == 0; }
static int g_attn_pv = 1;           // afd_debug_attn_rows 20 / 21: rank-8 products of the d = 8 kernels on the vector pipe (round 2) / on the fp16 matrix pipe
void attn_pv_set(int m) { g_attn_pv = m; }
void attn_mfma8_fwd(const float* qkv, float* o, float* lse, int B, int heads, int d, int L, float sc, hipStream_t s) {
  if (d == 8 && g_attn_pv) hipLaunchKernelGGL(attn_fwd_pv<8>, dim3(L / 256, heads, B), dim3(256), 0, s, qkv, o, lse, heads, L, sc);
  else if (d == 8) hipLaunchKernelGGL(attn_fwd_mfma<8>, dim3(L / 256, heads, B), dim3(256), 0, s, qkv, o, lse, heads, L, sc);
  else hipLaunchKernelGGL(attn_fwd_mfma<16>, dim3(L / 256, heads, B), dim3(256), 0, s, qkv, o, lse, heads, L, sc);
}
// dQ pass (+ delta) for d in {8, 16}; the dK / dV pass on these kernels only at d = 8: at d = 16 one key tile per wave is all
// the registers hold, every 16-wide Q / dO row read from LDS then serves one tile instead of two and the pass is LDS-bound
// (measured at sa1's shape: 138 us against 120 for the all-vector kernel with two key rows per lane, which stays)
void attn_mfma8_bwd_dq(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, float* delta,
                       int B, int heads, int d, int L, float sc, hipStream_t s) {
  if (d == 8) hipLaunchKernelGGL((attn_bwd_dq_mfma<8, 2>), dim3(L / 256, heads, B), dim3(256), 0, s, qkv, o, d_o, lse, dqkv, delta, heads, L, sc);
  else hipLaunchKernelGGL((attn_bwd_dq_mfma<16, 1>), dim3(L / 128, heads, B), dim3(256), 0, s, qkv, o, d_o, lse, dqkv, delta, heads, L, sc);
}
// the one-pass backward (d = 8): delta first, then the fused kernel -- one workgroup per (batch, head)
bool attn_fused_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, float* delta, int B, int heads,
                    int d, int L, float sc, hipStream_t s) {
  if (d != 8 || !g_attn_pv || L % 256) return false;
  const long total = (long)B * heads * L;
  hipLaunchKernelGGL(attn_delta_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, o, d_o, delta, heads, d, L, total);
  hipLaunchKernelGGL(attn_bwd_fused<8>, dim3(heads, B), dim3(256), 0, s, qkv, d_o, lse, delta, dqkv, heads, L, sc);
  return true;
}
void attn_mfma8_bwd_dkv(const float* qkv, const float* d_o, const float* lse, const float* delta, float* dqkv, int B, int heads, int L,
                        float sc, hipStream_t s) {
  hipLaunchKernelGGL((attn_bwd_dkv_mfma<8, 2>), dim3(L / 256, heads, B), dim3(256), 0, s, qkv, d_o, lse, delta, dqkv, heads, L, sc);
}

}  // namespace afd

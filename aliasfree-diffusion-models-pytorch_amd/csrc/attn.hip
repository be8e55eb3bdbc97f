// attn.hip -- F10 self-attention core: O = softmax(Q K^T / sqrt(d)) V per (batch, head), flash-style
// (online softmax, the L x L scores never exist in memory).
//
// Layout: qkv (B, 3C, L) = the NCHW image of the in-projection, channel = {q,k,v}*C + head*d + j,
// token index contiguous; o (B, C, L); lse (B, heads, L).
//
// Shapes here are d in {2,4,8,16,32,64}, L in {16..4096}.  With d = 8 an MFMA tile would run 50-75 %
// empty and the exp / max / rescale work is as large as the contractions, so these are VALU kernels:
// a lane owns R query (or key) rows -- their vectors, accumulators and running max / sum live in
// registers -- while the other operand streams through LDS in 64-row tiles stored [row][d], read with
// ONE ds_read_b128 per 4 values at a wave-uniform address (broadcast, conflict-free) and reused by
// all R rows of the lane.  History (profiles/): v1 read the tile with 16 ds_read_b32 per key and one
// row per lane -> LDS-issue bound (1.4 ms fwd / 3.5 ms bwd per step); v2 used wave-uniform scalar
// loads -> SGPR-bound backward (5.8 ms).  R rows per lane cut the LDS traffic per (q,k) pair by 4R.
// Forward: lane = query rows.  Backward: pass 1 lane = query rows (dQ, stores delta = rowsum(dO*O)),
// pass 2 lane = key rows (dK, dV) -- no atomics, bitwise reproducible.
#include "common.h"

namespace afd {

constexpr int kTile = 64;      // rows of the streamed operand per LDS tile
constexpr int kChunk = 8;      // keys per online-softmax rescale

// All d-length dot products and rank-d updates run on register PAIRS (v_pk_fma_f32 / v_pk_mul_f32, scalar operands
// broadcast through op_sel): measured on MI355X a packed FMA issues in 5.4 cycles against 4.3 for v_fma_f32
// (tools/micro/pk_rate.hip), i.e. 1.6x the FMA rate of the scalar form.  Scores live in the log2 domain
// (log2(e)/sqrt(d) folded into Q) so the exponential is one v_exp_f32.
using f2 = __attribute__((ext_vector_type(2))) float;
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

template <int D> struct RowVec {                 // one [d] row in LDS, read as b128 / b64 broadcasts into D/2 pairs
  static __device__ __forceinline__ void load(const float* __restrict__ p, f2 (&v)[D / 2]) {
    if constexpr (D % 4 == 0) {
#pragma unroll
      for (int j = 0; j < D / 4; ++j) {
        const float4 t = reinterpret_cast<const float4*>(p)[j];
        v[2 * j] = (f2){t.x, t.y}; v[2 * j + 1] = (f2){t.z, t.w};
      }
    } else {
      const float2 t = *reinterpret_cast<const float2*>(p);
      v[0] = (f2){t.x, t.y};
    }
  }
};
template <int P> __device__ __forceinline__ float dot2(const f2 (&a)[P], const f2 (&b)[P]) {
  f2 t = a[0] * b[0];
#pragma unroll
  for (int i = 1; i < P; ++i) t = __builtin_elementwise_fma(a[i], b[i], t);
  return t.x + t.y;
}
template <int P> __device__ __forceinline__ void axpy2(f2 (&acc)[P], float s, const f2 (&v)[P]) {
  const f2 ss = {s, s};
#pragma unroll
  for (int i = 0; i < P; ++i) acc[i] = __builtin_elementwise_fma(ss, v[i], acc[i]);
}

// stage rows [r0, r0+kTile) of a (D, L) j-major operand into LDS as [row][D] (optionally scaled).  NT = workgroup size:
// the trip count is a constant, so the loads of a tile are all in flight together (with a run-time bound each load
// waited for the previous store: at L <= 64, one wave per (batch, head), that latency chain WAS the kernel).
template <int D, int NT>
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, float* __restrict__ dst, int r0, int L, float mul) {
  constexpr int Q4 = D * kTile / 4;        // float4 slots of a tile
  if constexpr (Q4 % NT == 0) {
    if ((L & 3) == 0) {                     // 16-byte loads along the token index (r0 is a multiple of 64)
      constexpr int CAP = D >= 64 ? 2 : 8;                     // loads in flight (4 registers each; d = 64 has none to spare)
      constexpr int N4 = Q4 / NT, CH = N4 < CAP ? N4 : CAP;
#pragma unroll
      for (int e0 = 0; e0 < N4; e0 += CH) {
        float4 v[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          const int i = threadIdx.x + NT * (e0 + e), j = i / (kTile / 4), rr = 4 * (i % (kTile / 4));
          v[e] = (r0 + rr < L) ? *reinterpret_cast<const float4*>(src + (long)j * L + r0 + rr) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          const int i = threadIdx.x + NT * (e0 + e), j = i / (kTile / 4), rr = 4 * (i % (kTile / 4));
          dst[rr * D + j] = v[e].x * mul; dst[(rr + 1) * D + j] = v[e].y * mul;
          dst[(rr + 2) * D + j] = v[e].z * mul; dst[(rr + 3) * D + j] = v[e].w * mul;
        }
      }
      return;
    }
  }
#pragma unroll 4
  for (int i = threadIdx.x; i < D * kTile; i += NT) {
    const int j = i / kTile, rr = i % kTile;
    dst[rr * D + j] = (r0 + rr < L) ? src[(long)j * L + r0 + rr] * mul : 0.f;
  }
}

template <int D, int R, int NT>
__global__ __launch_bounds__(NT) void attn_fwd_k(const float* __restrict__ qkv, float* __restrict__ o,
                                                  float* __restrict__ lse, int heads, int L, float scale) {
  constexpr int P = D / 2;
  __shared__ __attribute__((aligned(16))) float Ks[kTile * D];
  __shared__ __attribute__((aligned(16))) float Vs[kTile * D];
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = heads * D;
  const int q0 = blockIdx.x * NT * R + threadIdx.x;           // row r of this lane = q0 + r*blockDim
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  f2 q[R][P], acc[R][P];
  float m[R], l[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qi = q0 + r * NT;
    m[r] = -INFINITY; l[r] = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) q[r][j >> 1][j & 1] = qi < L ? qp[(long)j * L + qi] * (scale * kLog2e) : 0.f;
#pragma unroll
    for (int i = 0; i < P; ++i) acc[r][i] = (f2){0.f, 0.f};
  }
  for (int k0 = 0; k0 < L; k0 += kTile) {
    __syncthreads();
    stage_rows<D, NT>(kp, Ks, k0, L, 1.f);
    stage_rows<D, NT>(vp, Vs, k0, L, 1.f);
    __syncthreads();
    const int nk = min(kTile, L - k0);
    for (int c0 = 0; c0 < nk; c0 += kChunk) {
      float s[R][kChunk], cm[R];
#pragma unroll
      for (int r = 0; r < R; ++r) cm[r] = m[r];
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        f2 kv[P];
        RowVec<D>::load(Ks + (c0 + u) * D, kv);
        const bool in = c0 + u < nk;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          s[r][u] = in ? dot2<P>(q[r], kv) : -INFINITY;
          cm[r] = fmaxf(cm[r], s[r][u]);
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float alpha = __builtin_amdgcn_exp2f(m[r] - cm[r]);      // m = -inf on the first chunk -> 0
        l[r] *= alpha;
#pragma unroll
        for (int i = 0; i < P; ++i) acc[r][i] *= alpha;
        m[r] = cm[r];
      }
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        f2 vv[P];
        RowVec<D>::load(Vs + (c0 + u) * D, vv);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[r][u] - cm[r]);
          l[r] += p;
          axpy2<P>(acc[r], p, vv);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qi = q0 + r * NT;
    if (qi < L) {
      const float inv = 1.0f / l[r];
      float* op = o + ((long)b * C + h * D) * L + qi;
#pragma unroll
      for (int j = 0; j < D; ++j) op[(long)j * L] = acc[r][j >> 1][j & 1] * inv;
      lse[((long)b * heads + h) * L + qi] = (m[r] + __builtin_amdgcn_logf(l[r])) * kLn2;      // v_log_f32 = log2
    }
  }
}

// dQ: lane = R query rows.  ds = p * (dp - delta) ; dq += ds * k * scale.  Also writes delta for the key pass.
template <int D, int R, int NT>
__global__ __launch_bounds__(NT) void attn_bwd_dq_k(const float* __restrict__ qkv, const float* __restrict__ o,
                                                     const float* __restrict__ d_o, const float* __restrict__ lse,
                                                     float* __restrict__ dqkv, float* __restrict__ delta_out,
                                                     int heads, int L, float scale) {
  constexpr int P = D / 2;
  __shared__ __attribute__((aligned(16))) float Ks[kTile * D];
  __shared__ __attribute__((aligned(16))) float Vs[kTile * D];
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = heads * D;
  const int q0 = blockIdx.x * NT * R + threadIdx.x;
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const long obase = ((long)b * C + h * D) * L;
  f2 q[R][P], go[R][P], dq[R][P];
  float delta[R], ls[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qi = q0 + r * NT;
    const bool live = qi < L;
    delta[r] = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float g = live ? d_o[obase + (long)j * L + qi] : 0.f;
      q[r][j >> 1][j & 1] = live ? qp[(long)j * L + qi] * (scale * kLog2e) : 0.f;
      go[r][j >> 1][j & 1] = g;
      delta[r] += g * (live ? o[obase + (long)j * L + qi] : 0.f);
    }
#pragma unroll
    for (int i = 0; i < P; ++i) dq[r][i] = (f2){0.f, 0.f};
    ls[r] = live ? lse[((long)b * heads + h) * L + qi] * kLog2e : 0.f;
    if (live) delta_out[((long)b * heads + h) * L + qi] = delta[r];
  }
  for (int k0 = 0; k0 < L; k0 += kTile) {
    __syncthreads();
    stage_rows<D, NT>(kp, Ks, k0, L, 1.f);
    stage_rows<D, NT>(vp, Vs, k0, L, 1.f);
    __syncthreads();
    const int nk = min(kTile, L - k0);
#pragma unroll 2
    for (int u = 0; u < nk; ++u) {
      f2 kv[P], vv[P];
      RowVec<D>::load(Ks + u * D, kv);
      RowVec<D>::load(Vs + u * D, vv);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float sc = dot2<P>(q[r], kv), dp = dot2<P>(go[r], vv);
        const float ds = __builtin_amdgcn_exp2f(sc - ls[r]) * (dp - delta[r]);
        axpy2<P>(dq[r], ds, kv);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qi = q0 + r * NT;
    if (qi < L) {
      float* dqp = dqkv + ((long)b * 3 * C + h * D) * L + qi;
#pragma unroll
      for (int j = 0; j < D; ++j) dqp[(long)j * L] = dq[r][j >> 1][j & 1] * scale;
    }
  }
}

// dK, dV: lane = R key rows; LDS holds a tile of (scale*log2e*Q), dO, lse*log2e and delta rows.
template <int D, int R, int NT>
__global__ __launch_bounds__(NT) void attn_bwd_dkv_k(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                      const float* __restrict__ lse, const float* __restrict__ delta,
                                                      float* __restrict__ dqkv, int heads, int L, float scale) {
  constexpr int P = D / 2;
  __shared__ __attribute__((aligned(16))) float Qs[kTile * D];
  __shared__ __attribute__((aligned(16))) float Gs[kTile * D];
  __shared__ float Ls[kTile];
  __shared__ float Ds[kTile];
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = heads * D;
  const int kq0 = blockIdx.x * NT * R + threadIdx.x;
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const float* gp = d_o + ((long)b * C + h * D) * L;
  const float* lp = lse + ((long)b * heads + h) * L;
  const float* dlp = delta + ((long)b * heads + h) * L;
  f2 k[R][P], v[R][P], dk[R][P], dv[R][P];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ki = kq0 + r * NT;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      k[r][j >> 1][j & 1] = ki < L ? kp[(long)j * L + ki] : 0.f;
      v[r][j >> 1][j & 1] = ki < L ? vp[(long)j * L + ki] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < P; ++i) { dk[r][i] = (f2){0.f, 0.f}; dv[r][i] = (f2){0.f, 0.f}; }
  }
  for (int t0 = 0; t0 < L; t0 += kTile) {
    __syncthreads();
    stage_rows<D, NT>(qp, Qs, t0, L, scale * kLog2e);
    stage_rows<D, NT>(gp, Gs, t0, L, 1.f);
    for (int i = threadIdx.x; i < kTile; i += NT) {
      const bool in = t0 + i < L;
      Ls[i] = in ? lp[t0 + i] * kLog2e : INFINITY;     // exp2(s - inf) = 0 for padded queries
      Ds[i] = in ? dlp[t0 + i] : 0.f;
    }
    __syncthreads();
    const int nq = min(kTile, L - t0);
#pragma unroll 2
    for (int u = 0; u < nq; ++u) {
      f2 qv[P], gv[P];
      RowVec<D>::load(Qs + u * D, qv);
      RowVec<D>::load(Gs + u * D, gv);
      const float lsu = Ls[u], dlu = Ds[u];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float sc = dot2<P>(qv, k[r]), dp = dot2<P>(gv, v[r]);
        const float p = __builtin_amdgcn_exp2f(sc - lsu);
        const float ds = p * (dp - dlu);
        axpy2<P>(dv[r], p, gv);
        axpy2<P>(dk[r], ds, qv);                       // qv carries scale*log2e; the log2e is taken out at the store
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ki = kq0 + r * NT;
    if (ki < L) {
      float* dkp = dqkv + ((long)b * 3 * C + C + h * D) * L + ki;
      float* dvp = dkp + (long)C * L;
#pragma unroll
      for (int j = 0; j < D; ++j) { dkp[(long)j * L] = dk[r][j >> 1][j & 1] * kLn2; dvp[(long)j * L] = dv[r][j >> 1][j & 1]; }
    }
  }
}

}  // namespace afd
using namespace afd;

template <int R> static inline void attn_geometry(int B, int heads, int L, dim3& grid, dim3& block) {
  const int rows = (L + R - 1) / R;
  block = dim3(rows <= 64 ? 64 : (rows <= 128 ? 128 : 256));
  grid = dim3((L + block.x * R - 1) / (block.x * R), heads, B);
}
template <int D, int R> static void launch_fwd(const float* qkv, float* o, float* lse, int B, int heads, int L, float sc, hipStream_t s) {
  dim3 grid, block; attn_geometry<R>(B, heads, L, grid, block);
  if (block.x == 64) hipLaunchKernelGGL((attn_fwd_k<D, R, 64>), grid, block, 0, s, qkv, o, lse, heads, L, sc);
  else if (block.x == 128) hipLaunchKernelGGL((attn_fwd_k<D, R, 128>), grid, block, 0, s, qkv, o, lse, heads, L, sc);
  else hipLaunchKernelGGL((attn_fwd_k<D, R, 256>), grid, block, 0, s, qkv, o, lse, heads, L, sc);
}
template <int D, int R> static void launch_dq(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                                              float* delta, int B, int heads, int L, float sc, hipStream_t s) {
  dim3 grid, block; attn_geometry<R>(B, heads, L, grid, block);
  if (block.x == 64) hipLaunchKernelGGL((attn_bwd_dq_k<D, R, 64>), grid, block, 0, s, qkv, o, d_o, lse, dqkv, delta, heads, L, sc);
  else if (block.x == 128) hipLaunchKernelGGL((attn_bwd_dq_k<D, R, 128>), grid, block, 0, s, qkv, o, d_o, lse, dqkv, delta, heads, L, sc);
  else hipLaunchKernelGGL((attn_bwd_dq_k<D, R, 256>), grid, block, 0, s, qkv, o, d_o, lse, dqkv, delta, heads, L, sc);
}
template <int D, int R> static void launch_dkv(const float* qkv, const float* d_o, const float* lse, const float* delta, float* dqkv,
                                               int B, int heads, int L, float sc, hipStream_t s) {
  dim3 grid, block; attn_geometry<R>(B, heads, L, grid, block);
  if (block.x == 64) hipLaunchKernelGGL((attn_bwd_dkv_k<D, R, 64>), grid, block, 0, s, qkv, d_o, lse, delta, dqkv, heads, L, sc);
  else if (block.x == 128) hipLaunchKernelGGL((attn_bwd_dkv_k<D, R, 128>), grid, block, 0, s, qkv, d_o, lse, delta, dqkv, heads, L, sc);
  else hipLaunchKernelGGL((attn_bwd_dkv_k<D, R, 256>), grid, block, 0, s, qkv, d_o, lse, delta, dqkv, heads, L, sc);
}

namespace afd {   // attn_mfma.hip: d = 8 / 16 passes with the d-contractions on the matrix cores
bool attn_mfma8_ok(int d, int L);
void attn_pv_set(int m);
bool attn_fused_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, float* delta, int B, int heads,
                    int d, int L, float sc, hipStream_t s);
void attn_mfma8_fwd(const float* qkv, float* o, float* lse, int B, int heads, int d, int L, float sc, hipStream_t s);
void attn_mfma8_bwd_dq(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, float* delta,
                       int B, int heads, int d, int L, float sc, hipStream_t s);
void attn_mfma8_bwd_dkv(const float* qkv, const float* d_o, const float* lse, const float* delta, float* dqkv, int B, int heads, int L,
                        float sc, hipStream_t s);
}
static int g_attn_mfma_bwd_all = 1;   // (afd_debug_attn_rows(10) off / (11) on): the MFMA d = 8 backward also at L = 1024 -- with the K / V tiles
                                      // prefetched through registers it beats the all-VALU pair there too (1.31 vs 1.49 ms at B = 256)
static int g_attn_rows = 0;      // tuning hook: 0 = default (MFMA path for d = 8 when L % 256 == 0); 1,2,4 force the
                                 // all-VALU kernels with that many rows per lane

extern "C" {

int afd_debug_attn_rows(int r) {
  if (r == 20 || r == 21) { attn_pv_set(r - 20); return AFD_OK; }                   // rank-8 products of the d = 8 MFMA kernels: vector pipe / fp16 matrix pipe (default)
  if (r == 10 || r == 11) { g_attn_mfma_bwd_all = r - 10; return AFD_OK; }         // MFMA d = 8 backward at L = 1024 off / on (default on)
  AFD_REQUIRE(r == 0 || r == 1 || r == 2 || r == 4, "afd_debug_attn_rows: r must be 0, 1, 2, 4 (rows per lane of the VALU kernels), 10, 11, 20 or 21");
  g_attn_rows = r;
  return AFD_OK;
}

int afd_attn_fwd(const float* qkv, float* o, float* lse, int B, int heads, int d, int L, afd_stream_t st) {
  AFD_REQUIRE(qkv && o && lse && B > 0 && heads > 0 && L > 0, "afd_attn_fwd: bad argument");
  AFD_REQUIRE(d == 2 || d == 4 || d == 8 || d == 16 || d == 32 || d == 64, "afd_attn_fwd: head dim %d not in {2,4,8,16,32,64}", d);
  AFD_REQUIRE(B <= 65535 && heads <= 65535, "afd_attn_fwd: grid too large");
  hipStream_t s = as_stream(st);
  const float sc = 1.0f / sqrtf((float)d);
  if (g_attn_rows == 0 && attn_mfma8_ok(d, L)) { attn_mfma8_fwd(qkv, o, lse, B, heads, d, L, sc, s); return check_launch("afd_attn_fwd"); }
  // rows per lane by head dim (register budget): 4,4,4,2,2,1
  switch (d) {
    case 2:  launch_fwd<2, 4>(qkv, o, lse, B, heads, L, sc, s); break;
    case 4:  launch_fwd<4, 4>(qkv, o, lse, B, heads, L, sc, s); break;
    case 8:
      // measured (tools/attn_bench.py, B=256): R=2 wins at L=1024, R=1 at L<=256 (4x the waves)
      if (g_attn_rows == 1 || (g_attn_rows == 0 && L < 1024)) launch_fwd<8, 1>(qkv, o, lse, B, heads, L, sc, s);
      else if (g_attn_rows == 4) launch_fwd<8, 4>(qkv, o, lse, B, heads, L, sc, s);
      else launch_fwd<8, 2>(qkv, o, lse, B, heads, L, sc, s);
      break;
    // L <= 64: one row per lane (with two, the second row of every lane would lie past the end of the sequence)
    case 16: if (L <= 64) launch_fwd<16, 1>(qkv, o, lse, B, heads, L, sc, s); else launch_fwd<16, 2>(qkv, o, lse, B, heads, L, sc, s); break;
    case 32: if (L <= 64) launch_fwd<32, 1>(qkv, o, lse, B, heads, L, sc, s); else launch_fwd<32, 2>(qkv, o, lse, B, heads, L, sc, s); break;
    default: launch_fwd<64, 1>(qkv, o, lse, B, heads, L, sc, s); break;
  }
  return check_launch("afd_attn_fwd");
}

int afd_attn_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, float* delta_ws,
                 int B, int heads, int d, int L, afd_stream_t st) {
  AFD_REQUIRE(qkv && o && d_o && lse && dqkv && delta_ws && B > 0 && heads > 0 && L > 0, "afd_attn_bwd: bad argument");
  AFD_REQUIRE(d == 2 || d == 4 || d == 8 || d == 16 || d == 32 || d == 64, "afd_attn_bwd: head dim %d not in {2,4,8,16,32,64}", d);
  AFD_REQUIRE(B <= 65535 && heads <= 65535, "afd_attn_bwd: grid too large");
  hipStream_t s = as_stream(st);
  const float sc = 1.0f / sqrtf((float)d);
  if (g_attn_rows == 0 && attn_mfma8_ok(d, L) && (L < 1024 || g_attn_mfma_bwd_all)) {
    if (attn_fused_bwd(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, d, L, sc, s)) return check_launch("afd_attn_bwd");
    attn_mfma8_bwd_dq(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, d, L, sc, s);
    if (d == 8) attn_mfma8_bwd_dkv(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s);
    else launch_dkv<16, 2>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s);       // (the d = 16 dK / dV pass stays on the vector kernel)
    return check_launch("afd_attn_bwd");
  }
  switch (d) {      // dQ pass: rows per lane 4,4,4,2,2,1
    case 2:  launch_dq<2, 4>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s); break;
    case 4:  launch_dq<4, 4>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s); break;
    case 8:
      if (g_attn_rows == 1 || (g_attn_rows == 0 && L < 1024)) launch_dq<8, 1>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s);
      else if (g_attn_rows == 4) launch_dq<8, 4>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s);
      else launch_dq<8, 2>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s);
      break;
    case 16: if (L <= 64) launch_dq<16, 1>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s); else launch_dq<16, 2>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s); break;
    case 32: if (L <= 64) launch_dq<32, 1>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s); else launch_dq<32, 2>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s); break;
    default: launch_dq<64, 1>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s); break;
  }
  switch (d) {      // dK/dV pass (4 vectors per row): 4,4,2,2,1,1
    case 2:  launch_dkv<2, 4>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s); break;
    case 4:  launch_dkv<4, 4>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s); break;
    case 8:
      if (g_attn_rows == 1 || (g_attn_rows == 0 && L < 1024)) launch_dkv<8, 1>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s);
      else if (g_attn_rows == 4) launch_dkv<8, 4>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s);
      else launch_dkv<8, 2>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s);
      break;
    case 16: if (L <= 64) launch_dkv<16, 1>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s); else launch_dkv<16, 2>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s); break;
    case 32: launch_dkv<32, 1>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s); break;
    default: launch_dkv<64, 1>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s); break;
  }
  return check_launch("afd_attn_bwd");
}

}  // extern "C"

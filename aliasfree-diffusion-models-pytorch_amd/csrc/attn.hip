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

template <int D> struct RowVec {                 // one [d] row in LDS, read as b128 / b64 broadcasts
  static __device__ __forceinline__ void load(const float* __restrict__ p, float (&v)[D]) {
    if constexpr (D % 4 == 0) {
#pragma unroll
      for (int j = 0; j < D / 4; ++j) {
        const float4 t = reinterpret_cast<const float4*>(p)[j];
        v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
      }
    } else {
      const float2 t = *reinterpret_cast<const float2*>(p);
      v[0] = t.x; v[1] = t.y;
    }
  }
};

// stage rows [r0, r0+kTile) of a (D, L) j-major operand into LDS as [row][D] (optionally scaled)
template <int D>
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, float* __restrict__ dst, int r0, int L, float mul) {
  for (int i = threadIdx.x; i < D * kTile; i += blockDim.x) {
    const int j = i / kTile, rr = i % kTile;
    dst[rr * D + j] = (r0 + rr < L) ? src[(long)j * L + r0 + rr] * mul : 0.f;
  }
}

template <int D, int R>
__global__ __launch_bounds__(256) void attn_fwd_k(const float* __restrict__ qkv, float* __restrict__ o,
                                                  float* __restrict__ lse, int heads, int L, float scale) {
  __shared__ __attribute__((aligned(16))) float Ks[kTile * D];
  __shared__ __attribute__((aligned(16))) float Vs[kTile * D];
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = heads * D;
  const int q0 = blockIdx.x * blockDim.x * R + threadIdx.x;           // row r of this lane = q0 + r*blockDim
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  float q[R][D], acc[R][D], m[R], l[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qi = q0 + r * blockDim.x;
    m[r] = -INFINITY; l[r] = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) { q[r][j] = qi < L ? qp[(long)j * L + qi] * scale : 0.f; acc[r][j] = 0.f; }
  }
  for (int k0 = 0; k0 < L; k0 += kTile) {
    __syncthreads();
    stage_rows<D>(kp, Ks, k0, L, 1.f);
    stage_rows<D>(vp, Vs, k0, L, 1.f);
    __syncthreads();
    const int nk = min(kTile, L - k0);
    for (int c0 = 0; c0 < nk; c0 += kChunk) {
      float s[R][kChunk], cm[R];
#pragma unroll
      for (int r = 0; r < R; ++r) cm[r] = m[r];
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        float kv[D];
        RowVec<D>::load(Ks + (c0 + u) * D, kv);
        const bool in = c0 + u < nk;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float a = 0.f;
#pragma unroll
          for (int j = 0; j < D; ++j) a += q[r][j] * kv[j];
          s[r][u] = in ? a : -INFINITY;
          cm[r] = fmaxf(cm[r], s[r][u]);
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float alpha = __expf(m[r] - cm[r]);      // m = -inf on the first chunk -> 0
        l[r] *= alpha;
#pragma unroll
        for (int j = 0; j < D; ++j) acc[r][j] *= alpha;
        m[r] = cm[r];
      }
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        float vv[D];
        RowVec<D>::load(Vs + (c0 + u) * D, vv);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float p = __expf(s[r][u] - cm[r]);
          l[r] += p;
#pragma unroll
          for (int j = 0; j < D; ++j) acc[r][j] += p * vv[j];
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qi = q0 + r * blockDim.x;
    if (qi < L) {
      const float inv = 1.0f / l[r];
      float* op = o + ((long)b * C + h * D) * L + qi;
#pragma unroll
      for (int j = 0; j < D; ++j) op[(long)j * L] = acc[r][j] * inv;
      lse[((long)b * heads + h) * L + qi] = m[r] + __logf(l[r]);
    }
  }
}

// dQ: lane = R query rows.  ds = p * (dp - delta) ; dq += ds * k * scale.  Also writes delta for the key pass.
template <int D, int R>
__global__ __launch_bounds__(256) void attn_bwd_dq_k(const float* __restrict__ qkv, const float* __restrict__ o,
                                                     const float* __restrict__ d_o, const float* __restrict__ lse,
                                                     float* __restrict__ dqkv, float* __restrict__ delta_out,
                                                     int heads, int L, float scale) {
  __shared__ __attribute__((aligned(16))) float Ks[kTile * D];
  __shared__ __attribute__((aligned(16))) float Vs[kTile * D];
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = heads * D;
  const int q0 = blockIdx.x * blockDim.x * R + threadIdx.x;
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const long obase = ((long)b * C + h * D) * L;
  float q[R][D], go[R][D], dq[R][D], delta[R], ls[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qi = q0 + r * blockDim.x;
    const bool live = qi < L;
    delta[r] = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      q[r][j] = live ? qp[(long)j * L + qi] * scale : 0.f;
      go[r][j] = live ? d_o[obase + (long)j * L + qi] : 0.f;
      delta[r] += go[r][j] * (live ? o[obase + (long)j * L + qi] : 0.f);
      dq[r][j] = 0.f;
    }
    ls[r] = live ? lse[((long)b * heads + h) * L + qi] : 0.f;
    if (live) delta_out[((long)b * heads + h) * L + qi] = delta[r];
  }
  for (int k0 = 0; k0 < L; k0 += kTile) {
    __syncthreads();
    stage_rows<D>(kp, Ks, k0, L, 1.f);
    stage_rows<D>(vp, Vs, k0, L, 1.f);
    __syncthreads();
    const int nk = min(kTile, L - k0);
#pragma unroll 2
    for (int u = 0; u < nk; ++u) {
      float kv[D], vv[D];
      RowVec<D>::load(Ks + u * D, kv);
      RowVec<D>::load(Vs + u * D, vv);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float sc = 0.f, dp = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) { sc += q[r][j] * kv[j]; dp += go[r][j] * vv[j]; }
        const float ds = __expf(sc - ls[r]) * (dp - delta[r]);
#pragma unroll
        for (int j = 0; j < D; ++j) dq[r][j] += ds * kv[j];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qi = q0 + r * blockDim.x;
    if (qi < L) {
      float* dqp = dqkv + ((long)b * 3 * C + h * D) * L + qi;
#pragma unroll
      for (int j = 0; j < D; ++j) dqp[(long)j * L] = dq[r][j] * scale;
    }
  }
}

// dK, dV: lane = R key rows; LDS holds a tile of (scale*Q), dO, lse and delta rows.
template <int D, int R>
__global__ __launch_bounds__(256) void attn_bwd_dkv_k(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                      const float* __restrict__ lse, const float* __restrict__ delta,
                                                      float* __restrict__ dqkv, int heads, int L, float scale) {
  __shared__ __attribute__((aligned(16))) float Qs[kTile * D];
  __shared__ __attribute__((aligned(16))) float Gs[kTile * D];
  __shared__ float Ls[kTile];
  __shared__ float Ds[kTile];
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = heads * D;
  const int kq0 = blockIdx.x * blockDim.x * R + threadIdx.x;
  const float* qp = qkv + ((long)b * 3 * C + h * D) * L;
  const float* kp = qp + (long)C * L;
  const float* vp = kp + (long)C * L;
  const float* gp = d_o + ((long)b * C + h * D) * L;
  const float* lp = lse + ((long)b * heads + h) * L;
  const float* dlp = delta + ((long)b * heads + h) * L;
  float k[R][D], v[R][D], dk[R][D], dv[R][D];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ki = kq0 + r * blockDim.x;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      k[r][j] = ki < L ? kp[(long)j * L + ki] : 0.f;
      v[r][j] = ki < L ? vp[(long)j * L + ki] : 0.f;
      dk[r][j] = 0.f; dv[r][j] = 0.f;
    }
  }
  for (int t0 = 0; t0 < L; t0 += kTile) {
    __syncthreads();
    stage_rows<D>(qp, Qs, t0, L, scale);
    stage_rows<D>(gp, Gs, t0, L, 1.f);
    for (int i = threadIdx.x; i < kTile; i += blockDim.x) {
      const bool in = t0 + i < L;
      Ls[i] = in ? lp[t0 + i] : INFINITY;              // exp(s - inf) = 0 for padded queries
      Ds[i] = in ? dlp[t0 + i] : 0.f;
    }
    __syncthreads();
    const int nq = min(kTile, L - t0);
#pragma unroll 2
    for (int u = 0; u < nq; ++u) {
      float qv[D], gv[D];
      RowVec<D>::load(Qs + u * D, qv);
      RowVec<D>::load(Gs + u * D, gv);
      const float lsu = Ls[u], dlu = Ds[u];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float sc = 0.f, dp = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) { sc += qv[j] * k[r][j]; dp += gv[j] * v[r][j]; }
        const float p = __expf(sc - lsu);
        const float ds = p * (dp - dlu);
#pragma unroll
        for (int j = 0; j < D; ++j) { dv[r][j] += p * gv[j]; dk[r][j] += ds * qv[j]; }   // qv already carries `scale`
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ki = kq0 + r * blockDim.x;
    if (ki < L) {
      float* dkp = dqkv + ((long)b * 3 * C + C + h * D) * L + ki;
      float* dvp = dkp + (long)C * L;
#pragma unroll
      for (int j = 0; j < D; ++j) { dkp[(long)j * L] = dk[r][j]; dvp[(long)j * L] = dv[r][j]; }
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
  hipLaunchKernelGGL((attn_fwd_k<D, R>), grid, block, 0, s, qkv, o, lse, heads, L, sc);
}
template <int D, int R> static void launch_dq(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                                              float* delta, int B, int heads, int L, float sc, hipStream_t s) {
  dim3 grid, block; attn_geometry<R>(B, heads, L, grid, block);
  hipLaunchKernelGGL((attn_bwd_dq_k<D, R>), grid, block, 0, s, qkv, o, d_o, lse, dqkv, delta, heads, L, sc);
}
template <int D, int R> static void launch_dkv(const float* qkv, const float* d_o, const float* lse, const float* delta, float* dqkv,
                                               int B, int heads, int L, float sc, hipStream_t s) {
  dim3 grid, block; attn_geometry<R>(B, heads, L, grid, block);
  hipLaunchKernelGGL((attn_bwd_dkv_k<D, R>), grid, block, 0, s, qkv, d_o, lse, delta, dqkv, heads, L, sc);
}

static int g_attn_rows = 0;      // tuning hook: 0 = default rows per lane, else force R in {1,2,4} for d = 8

extern "C" {

int afd_debug_attn_rows(int r) {
  AFD_REQUIRE(r == 0 || r == 1 || r == 2 || r == 4, "afd_debug_attn_rows: r must be 0, 1, 2 or 4");
  g_attn_rows = r;
  return AFD_OK;
}

int afd_attn_fwd(const float* qkv, float* o, float* lse, int B, int heads, int d, int L, afd_stream_t st) {
  AFD_REQUIRE(qkv && o && lse && B > 0 && heads > 0 && L > 0, "afd_attn_fwd: bad argument");
  AFD_REQUIRE(d == 2 || d == 4 || d == 8 || d == 16 || d == 32 || d == 64, "afd_attn_fwd: head dim %d not in {2,4,8,16,32,64}", d);
  AFD_REQUIRE(B <= 65535 && heads <= 65535, "afd_attn_fwd: grid too large");
  hipStream_t s = as_stream(st);
  const float sc = 1.0f / sqrtf((float)d);
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
    case 16: launch_fwd<16, 2>(qkv, o, lse, B, heads, L, sc, s); break;
    case 32: launch_fwd<32, 2>(qkv, o, lse, B, heads, L, sc, s); break;
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
  switch (d) {      // dQ pass: rows per lane 4,4,4,2,2,1
    case 2:  launch_dq<2, 4>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s); break;
    case 4:  launch_dq<4, 4>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s); break;
    case 8:
      if (g_attn_rows == 1 || (g_attn_rows == 0 && L < 1024)) launch_dq<8, 1>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s);
      else if (g_attn_rows == 4) launch_dq<8, 4>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s);
      else launch_dq<8, 2>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s);
      break;
    case 16: launch_dq<16, 2>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s); break;
    case 32: launch_dq<32, 2>(qkv, o, d_o, lse, dqkv, delta_ws, B, heads, L, sc, s); break;
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
    case 16: launch_dkv<16, 2>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s); break;
    case 32: launch_dkv<32, 1>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s); break;
    default: launch_dkv<64, 1>(qkv, d_o, lse, delta_ws, dqkv, B, heads, L, sc, s); break;
  }
  return check_launch("afd_attn_bwd");
}

}  // extern "C"

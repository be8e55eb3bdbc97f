// elementwise.hip -- small HBM-bound helpers: GELU, 2x2 max-pool, bilinear 2x (align_corners),
// batched strided copy (virtual concat), add, column sums, sinusoidal time encoding, SiLU->Linear.
#include "common.h"

namespace afd {

static inline int gs_grid(long total, int block = 256) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 32768 ? 32768 : g));
}

#define AFD_GRID_STRIDE(i, total) \
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (total); i += (long)gridDim.x * blockDim.x)

// ---- GELU (nn.GELU exact) -----------------------------------------------------------------
__global__ void gelu_fwd_k(const float* __restrict__ x, float* __restrict__ y, long n4, long n) {
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float4* y4 = reinterpret_cast<float4*>(y);
  AFD_GRID_STRIDE(i, n4) {
    float4 v = x4[i];
    v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
    y4[i] = v;
  }
  const long tail0 = n4 * 4;
  AFD_GRID_STRIDE(i, n - tail0) y[tail0 + i] = gelu_erf(x[tail0 + i]);
}
__global__ void gelu_bwd_k(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n4, long n) {
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* g4 = reinterpret_cast<const float4*>(dy);
  float4* o4 = reinterpret_cast<float4*>(dx);
  AFD_GRID_STRIDE(i, n4) {
    const float4 v = x4[i], g = g4[i];
    o4[i] = make_float4(g.x * gelu_erf_grad(v.x), g.y * gelu_erf_grad(v.y), g.z * gelu_erf_grad(v.z), g.w * gelu_erf_grad(v.w));
  }
  const long tail0 = n4 * 4;
  AFD_GRID_STRIDE(i, n - tail0) dx[tail0 + i] = dy[tail0 + i] * gelu_erf_grad(x[tail0 + i]);
}

// ---- MaxPool2d(2): floor mode, first maximum wins in backward (ATen scan order) ----------
__global__ void maxpool2_fwd_k(const float* __restrict__ x, float* __restrict__ y, int H, int W, long total) {
  const int Ho = H / 2, Wo = W / 2;
  AFD_GRID_STRIDE(i, total) {
    const int j = i % Wo, r = (i / Wo) % Ho; const long pl = i / ((long)Wo * Ho);
    const float* p = x + pl * H * W + (long)(2 * r) * W + 2 * j;
    y[i] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[W], p[W + 1]));
  }
}
__global__ void maxpool2_bwd_k(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int H, int W, long total) {
  // one thread per INPUT element; odd trailing row/column receive zero
  const int Ho = H / 2, Wo = W / 2;
  AFD_GRID_STRIDE(i, total) {
    const int q = i % W, p = (i / W) % H; const long pl = i / ((long)W * H);
    const int r = p >> 1, j = q >> 1;
    float g = 0.f;
    if (r < Ho && j < Wo) {
      const float* w = x + pl * H * W + (long)(2 * r) * W + 2 * j;
      const float v[4] = {w[0], w[1], w[W], w[W + 1]};
      int arg = 0; float m = v[0];
#pragma unroll
      for (int k = 1; k < 4; ++k) if (v[k] > m || (v[k] != v[k] && m == m)) { m = v[k]; arg = k; }
      if (arg == (p & 1) * 2 + (q & 1)) g = dy[pl * Ho * Wo + (long)r * Wo + j];
    }
    dx[i] = g;
  }
}

// ---- Upsample(scale_factor=2, bilinear, align_corners=True) -------------------------------
// src = dst * (H-1)/(2H-1) (ATen area_pixel_compute_scale with align_corners); fp32 like ATen.
__device__ __forceinline__ void bil_src(int d, int n_in, float scale, int& i0, int& i1, float& l1) {
  const float s = scale * d;
  i0 = (int)s; if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = s - i0;
}
__global__ void bilinear_up2_fwd_k(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W, long ybs, long total) {
  const int H2 = 2 * H, W2 = 2 * W;
  const float sh = H2 > 1 ? (float)(H - 1) / (H2 - 1) : 0.f, sw = W2 > 1 ? (float)(W - 1) / (W2 - 1) : 0.f;
  AFD_GRID_STRIDE(i, total) {
    const int q = i % W2, p = (i / W2) % H2; const long bc = i / ((long)W2 * H2);
    const int c = bc % C; const long b = bc / C;
    int y0, y1, x0, x1; float ly, lx;
    bil_src(p, H, sh, y0, y1, ly); bil_src(q, W, sw, x0, x1, lx);
    const float* s = x + bc * H * W;
    const float hy = 1.f - ly, hx = 1.f - lx;
    y[b * ybs + (long)c * H2 * W2 + (long)p * W2 + q] =
        hy * (hx * s[y0 * W + x0] + lx * s[y0 * W + x1]) + ly * (hx * s[y1 * W + x0] + lx * s[y1 * W + x1]);
  }
}
// gather form of the adjoint: each input pixel sums the <= 3x3 output pixels that reference it
__global__ void bilinear_up2_bwd_k(const float* __restrict__ dy, float* __restrict__ dx, int C, int H, int W, long dybs, long total) {
  const int H2 = 2 * H, W2 = 2 * W;
  const float sh = H2 > 1 ? (float)(H - 1) / (H2 - 1) : 0.f, sw = W2 > 1 ? (float)(W - 1) / (W2 - 1) : 0.f;
  AFD_GRID_STRIDE(i, total) {
    const int j = i % W, r = (i / W) % H; const long bc = i / ((long)W * H);
    const int c = bc % C; const long b = bc / C;
    const float* g = dy + b * dybs + (long)c * H2 * W2;
    // candidate output rows: those whose y0 or y1 can equal r  -> p in [2r-2, 2r+3]
    float acc = 0.f;
    for (int p = max(0, 2 * r - 2); p <= min(H2 - 1, 2 * r + 3); ++p) {
      int y0, y1; float ly; bil_src(p, H, sh, y0, y1, ly);
      float wy = 0.f;
      if (y0 == r) wy += 1.f - ly;
      if (y1 == r) wy += ly;
      if (wy == 0.f) continue;
      for (int q = max(0, 2 * j - 2); q <= min(W2 - 1, 2 * j + 3); ++q) {
        int x0, x1; float lx; bil_src(q, W, sw, x0, x1, lx);
        float wx = 0.f;
        if (x0 == j) wx += 1.f - lx;
        if (x1 == j) wx += lx;
        if (wx != 0.f) acc += wy * wx * g[(long)p * W2 + q];
      }
    }
    dx[i] = acc;
  }
}

// ---- batched strided copy, add, column sums -----------------------------------------------
__global__ void copy_batched_k(const float* __restrict__ s, float* __restrict__ d, long n, long sbs, long dbs, long total) {
  AFD_GRID_STRIDE(i, total) { const long b = i / n, k = i % n; d[b * dbs + k] = s[b * sbs + k]; }
}
__global__ void add_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n) {
  AFD_GRID_STRIDE(i, n) y[i] = a[i] + b[i];
}
// out[j] = sum_i in[i*cols + j].  256 threads = 32 columns x 8 row groups; fixed order (deterministic)
__global__ __launch_bounds__(256) void colsum_k(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, long rs, int accumulate) {
  __shared__ float red[8][33];
  const int j = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
  float s0 = 0.f, s1 = 0.f;
  if (j < cols) {
    int i = g;
    for (; i + 8 < rows; i += 16) { s0 += in[(long)i * rs + j]; s1 += in[(long)(i + 8) * rs + j]; }
    if (i < rows) s0 += in[(long)i * rs + j];
  }
  red[g][threadIdx.x & 31] = s0 + s1;
  __syncthreads();
  if (g == 0 && j < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31];
    out[j] = accumulate ? out[j] + t : t;
  }
}

// ---- time embedding ------------------------------------------------------------------------
__global__ void pos_encoding_k(const int64_t* __restrict__ t, const float* __restrict__ inv_freq, float* __restrict__ temb, int B, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i % half;
  const float a = (float)t[b] * inv_freq[k];       // t.float() * inv_freq, one fp32 product (ddpm_models.py:266)
  temb[(long)b * 2 * half + k] = sinf(a);
  temb[(long)b * 2 * half + half + k] = cosf(a);
}

__device__ __forceinline__ float silu(float v) { return v / (1.f + __expf(-v)); }
__device__ __forceinline__ float silu_grad(float v) { const float s = 1.f / (1.f + __expf(-v)); return s * (1.f + v * (1.f - s)); }

// one wave per output (b, n): lanes stride over K
__global__ void silu_linear_fwd_k(const float* __restrict__ temb, const float* __restrict__ w, const float* __restrict__ bias,
                                  float* __restrict__ out, int B, int K, int N) {
  const long wv = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wv >= (long)B * N) return;
  const int b = wv / N, n = wv % N;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += silu(temb[(long)b * K + k]) * w[(long)n * K + k];
  s = wave_sum(s);
  if (lane == 0) out[wv] = s + (bias ? bias[n] : 0.f);
}
// the same for up to 8 layers that share temb (the six stages' emb_layer of one UNet forward: their input exists as soon as
// the forward starts, so ONE launch replaces six small dependent ones); a wave finds its layer by the output offsets
struct SiluBatch {
  const float* w[8]; const float* bias[8]; float* out[8];
  int N[8], first[9];            // outputs per row; first output column of layer i in the concatenated row
  int n;
};
// a wave = one batch row x EIGHT consecutive outputs of the concatenated row (the SiLU of the row's K inputs is evaluated once
// per lane and feeds eight dot products; DPP wave sums): 45 -> ~10 us for the UNet's six layers at B = 256
__global__ __launch_bounds__(256) void silu_linear_fwd_batched_k(const float* __restrict__ temb, SiluBatch d, int B, int K) {
  const int NT = d.first[d.n], G = (NT + 7) >> 3;                      // output groups of 8 per row
  const long wv = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wv >= (long)B * G) return;
  const int b = wv / G, c0 = (wv % G) * 8;
  float s[8];
  const float* wp[8];
  int li[8], ni[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int col = c0 + j < NT ? c0 + j : NT - 1;                     // (a ragged last group repeats the last column; not stored)
    int l = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) if (q < d.n && col >= d.first[q]) l = q;
    li[j] = l; ni[j] = col - d.first[l];
    wp[j] = d.w[l] + (long)ni[j] * K;
    s[j] = 0.f;
  }
  for (int k = lane; k < K; k += 64) {
    const float a = silu(temb[(long)b * K + k]);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = fmaf(a, wp[j][k], s[j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float t = wave_sum_to_lane63(s[j]);
    if (lane == 63 && c0 + j < NT) d.out[li[j]][(long)b * d.N[li[j]] + ni[j]] = t + (d.bias[li[j]] ? d.bias[li[j]][ni[j]] : 0.f);
  }
}
// out_i[b][:] = table_i[clamp(idx[b])][:] for up to 8 tables that share idx (the six stages' time-embedding tables of a
// sampling run: every timestep's emb_layer output is computed once per trajectory, a denoise step only gathers its rows)
__global__ void gather_rows_batched_k(const int64_t* __restrict__ idx, SiluBatch d, int B, int rows) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int NT = d.first[d.n];
  if (i >= (long)B * NT) return;
  const int b = i / NT, col = i % NT;
  int l = 0;
#pragma unroll
  for (int j = 1; j < 8; ++j) if (j < d.n && col >= d.first[j]) l = j;
  const int n = col - d.first[l];
  long r = idx[b];
  r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
  d.out[l][(long)b * d.N[l] + n] = d.w[l][r * d.N[l] + n];
}
// dw[n,k] = sum_b dout[b,n] silu(temb[b,k]).  Block = EIGHT output rows n x 32 columns k x 8 batch groups: a thread evaluates
// silu(temb[b,k]) once per batch row and feeds eight running sums (the eight dout values of a row are one broadcast 32-byte
// read); the 8 batch groups' partial sums meet in LDS (fixed order).  (One row per block, the first form, spent 20 us per
// stage on 2-8 MFLOP: 32 dependent iterations of two loads and one FMA.)  Row block blockIdx.y == ceil(N / 8) (present when
// db != NULL) holds the bias gradient db[n] = sum_b dout[b,n] instead: 32 outputs n per block -- one launch for both.
__global__ __launch_bounds__(256) void silu_linear_dw_k(const float* __restrict__ temb, const float* __restrict__ dout, float* __restrict__ dw,
                                                        float* __restrict__ db, int B, int K, int N, int accumulate) {
  __shared__ float red[8][8][33];
  const int g = threadIdx.x >> 5, l = threadIdx.x & 31;
  const int nrb = (N + 7) >> 3;
  if ((int)blockIdx.y == nrb) {
    const int n = blockIdx.x * 32 + l;
    if (blockIdx.x * 32 >= N) return;
    float s0 = 0.f, s1 = 0.f;
    if (n < N) {
      int b = g;
      for (; b + 8 < B; b += 16) { s0 += dout[(long)b * N + n]; s1 += dout[(long)(b + 8) * N + n]; }
      if (b < B) s0 += dout[(long)b * N + n];
    }
    red[0][g][l] = s0 + s1;
    __syncthreads();
    if (g == 0 && n < N) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) t += red[0][j][l];
      db[n] = accumulate ? db[n] + t : t;
    }
    return;
  }
  const int n0 = blockIdx.y * 8;
  const int k = blockIdx.x * 32 + l;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  if (k < K) {
    int b = g;
    for (; b + 24 < B; b += 32) {                                      // four batch rows in flight
      float a[4], dv[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = temb[(long)(b + 8 * u) * K + k];
        const float* d = dout + (long)(b + 8 * u) * N + n0;
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[u][j] = n0 + j < N ? d[j] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float av = silu(a[u]);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = fmaf(dv[u][j], av, s[j]);
      }
    }
    for (; b < B; b += 8) {
      const float av = silu(temb[(long)b * K + k]);
      const float* d = dout + (long)b * N + n0;
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] = fmaf(n0 + j < N ? d[j] : 0.f, av, s[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[j][g][l] = s[j];
  __syncthreads();
  // thread (j = g, l): output row n0 + g, column k
  if (k < K && n0 + g < N) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[g][q][l];
    float* o = dw + (long)(n0 + g) * K + k;
    *o = accumulate ? *o + t : t;
  }
}
// dtemb[b,k] += silu'(temb[b,k]) * sum_n dout[b,n] w[n,k]
__global__ void silu_linear_dx_k(const float* __restrict__ temb, const float* __restrict__ w, const float* __restrict__ dout,
                                 float* __restrict__ dtemb, int B, int K, int N) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * K) return;
  const int b = i / K, k = i % K;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += dout[(long)b * N + n] * w[(long)n * K + k];
  dtemb[i] += s * silu_grad(temb[i]);
}

}  // namespace afd
using namespace afd;

extern "C" {

int afd_gelu_fwd(const float* x, float* y, long n, afd_stream_t st) {
  AFD_REQUIRE(x && y && n > 0, "afd_gelu_fwd: bad argument");
  const bool al = (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
  const long n4 = al ? n / 4 : 0;
  hipLaunchKernelGGL(gelu_fwd_k, dim3(gs_grid(n4 ? n4 : n)), dim3(256), 0, as_stream(st), x, y, n4, n);
  return check_launch("afd_gelu_fwd");
}
int afd_gelu_bwd(const float* x, const float* dy, float* dx, long n, afd_stream_t st) {
  AFD_REQUIRE(x && dy && dx && n > 0, "afd_gelu_bwd: bad argument");
  const bool al = (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
  const long n4 = al ? n / 4 : 0;
  hipLaunchKernelGGL(gelu_bwd_k, dim3(gs_grid(n4 ? n4 : n)), dim3(256), 0, as_stream(st), x, dy, dx, n4, n);
  return check_launch("afd_gelu_bwd");
}
int afd_maxpool2_fwd(const float* x, float* y, int B, int C, int H, int W, afd_stream_t st) {
  AFD_REQUIRE(x && y && B > 0 && C > 0 && H >= 2 && W >= 2, "afd_maxpool2_fwd: bad argument");
  const long total = (long)B * C * (H / 2) * (W / 2);
  hipLaunchKernelGGL(maxpool2_fwd_k, dim3(gs_grid(total)), dim3(256), 0, as_stream(st), x, y, H, W, total);
  return check_launch("afd_maxpool2_fwd");
}
int afd_maxpool2_bwd(const float* x, const float* dy, float* dx, int B, int C, int H, int W, afd_stream_t st) {
  AFD_REQUIRE(x && dy && dx && B > 0 && C > 0 && H >= 2 && W >= 2, "afd_maxpool2_bwd: bad argument");
  const long total = (long)B * C * H * W;
  hipLaunchKernelGGL(maxpool2_bwd_k, dim3(gs_grid(total)), dim3(256), 0, as_stream(st), x, dy, dx, H, W, total);
  return check_launch("afd_maxpool2_bwd");
}
int afd_bilinear_up2_fwd(const float* x, float* y, int B, int C, int H, int W, long ybs, afd_stream_t st) {
  AFD_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "afd_bilinear_up2_fwd: bad argument");
  if (!ybs) ybs = (long)C * 4 * H * W;
  const long total = (long)B * C * 4 * H * W;
  hipLaunchKernelGGL(bilinear_up2_fwd_k, dim3(gs_grid(total)), dim3(256), 0, as_stream(st), x, y, C, H, W, ybs, total);
  return check_launch("afd_bilinear_up2_fwd");
}
int afd_bilinear_up2_bwd(const float* dy, float* dx, int B, int C, int H, int W, long dybs, afd_stream_t st) {
  AFD_REQUIRE(dy && dx && B > 0 && C > 0 && H > 0 && W > 0, "afd_bilinear_up2_bwd: bad argument");
  if (!dybs) dybs = (long)C * 4 * H * W;
  const long total = (long)B * C * H * W;
  hipLaunchKernelGGL(bilinear_up2_bwd_k, dim3(gs_grid(total)), dim3(256), 0, as_stream(st), dy, dx, C, H, W, dybs, total);
  return check_launch("afd_bilinear_up2_bwd");
}
int afd_copy_batched(const float* src, float* dst, int B, long n, long sbs, long dbs, afd_stream_t st) {
  AFD_REQUIRE(src && dst && B > 0 && n > 0, "afd_copy_batched: bad argument");
  if (!sbs) sbs = n;
  if (!dbs) dbs = n;
  const long total = (long)B * n;
  hipLaunchKernelGGL(copy_batched_k, dim3(gs_grid(total)), dim3(256), 0, as_stream(st), src, dst, n, sbs, dbs, total);
  return check_launch("afd_copy_batched");
}
int afd_add(const float* a, const float* b, float* y, long n, afd_stream_t st) {
  AFD_REQUIRE(a && b && y && n > 0, "afd_add: bad argument");
  hipLaunchKernelGGL(add_k, dim3(gs_grid(n)), dim3(256), 0, as_stream(st), a, b, y, n);
  return check_launch("afd_add");
}
int afd_colsum(const float* in, float* out, int rows, int cols, int accumulate, afd_stream_t st) {
  AFD_REQUIRE(in && out && rows > 0 && cols > 0, "afd_colsum: bad argument");
  hipLaunchKernelGGL(colsum_k, dim3((cols + 31) / 32), dim3(256), 0, as_stream(st), in, out, rows, cols, (long)cols, accumulate);
  return check_launch("afd_colsum");
}
// (rows, 2, C) partials -> out_a[C] (+)= column sums of block 0, out_b[C] (+)= block 1; ONE launch
__global__ __launch_bounds__(256) void colsum2_k(const float* __restrict__ in, float* __restrict__ oa, float* __restrict__ ob,
                                                 int rows, int C, int accumulate) {
  __shared__ float red[8][33];
  const int j = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;     // j in [0, 2C)
  float s0 = 0.f, s1 = 0.f;
  if (j < 2 * C) {
    int i = g;
    for (; i + 8 < rows; i += 16) { s0 += in[(long)i * 2 * C + j]; s1 += in[(long)(i + 8) * 2 * C + j]; }
    if (i < rows) s0 += in[(long)i * 2 * C + j];
  }
  red[g][threadIdx.x & 31] = s0 + s1;
  __syncthreads();
  if (g == 0 && j < 2 * C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31];
    float* o = j < C ? oa + j : ob + (j - C);
    *o = accumulate ? *o + t : t;
  }
}

int afd_colsum_strided(const float* in, long row_stride, float* out, int rows, int cols, int accumulate, afd_stream_t st) {
  AFD_REQUIRE(in && out && rows > 0 && cols > 0 && row_stride >= cols, "afd_colsum_strided: bad argument");
  hipLaunchKernelGGL(colsum_k, dim3((cols + 31) / 32), dim3(256), 0, as_stream(st), in, out, rows, cols, row_stride, accumulate);
  return check_launch("afd_colsum_strided");
}
int afd_colsum2(const float* in, float* out_a, float* out_b, int rows, int C, int accumulate, afd_stream_t st) {
  AFD_REQUIRE(in && out_a && out_b && rows > 0 && C > 0, "afd_colsum2: bad argument");
  hipLaunchKernelGGL(colsum2_k, dim3((2 * C + 31) / 32), dim3(256), 0, as_stream(st), in, out_a, out_b, rows, C, accumulate);
  return check_launch("afd_colsum2");
}
// label conditioning (ddpm_models.py:254,276-277): t_emb += label_emb(y)
__global__ __launch_bounds__(256) void embed_add_fwd_k(const float* __restrict__ temb, const float* __restrict__ table,
                                                       const int64_t* __restrict__ y, float* __restrict__ out, int B, int D,
                                                       int K) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  long k = y[b];
  k = k < 0 ? 0 : (k >= K ? K - 1 : k);          // an out-of-range label cannot fault the launch (torch would raise)
  out[i] = temb[i] + table[k * D + d];
}
// one thread per (class, column); the batch is walked in order, so the row sums are deterministic
__global__ __launch_bounds__(256) void embed_add_bwd_k(const float* __restrict__ dout, const int64_t* __restrict__ y,
                                                       float* __restrict__ dtable, int B, int D, int K, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K * D) return;
  const int k = i / D, d = i - k * D;
  float s = 0.f;
  for (int b = 0; b < B; ++b)
    if (y[b] == k) s += dout[(long)b * D + d];
  dtable[i] = accumulate ? dtable[i] + s : s;
}

int afd_embed_add_fwd(const float* temb, const float* table, const int64_t* y, float* out, int B, int D, int num_classes,
                      afd_stream_t st) {
  AFD_REQUIRE(temb && table && y && out && B > 0 && D > 0 && num_classes > 0, "afd_embed_add_fwd: bad argument");
  hipLaunchKernelGGL(embed_add_fwd_k, dim3((B * D + 255) / 256), dim3(256), 0, as_stream(st), temb, table, y, out, B, D, num_classes);
  return check_launch("afd_embed_add_fwd");
}
int afd_embed_add_bwd(const float* dout, const int64_t* y, float* dtable, int B, int D, int num_classes, int accumulate,
                      afd_stream_t st) {
  AFD_REQUIRE(dout && y && dtable && B > 0 && D > 0 && num_classes > 0, "afd_embed_add_bwd: bad argument");
  hipLaunchKernelGGL(embed_add_bwd_k, dim3((num_classes * D + 255) / 256), dim3(256), 0, as_stream(st), dout, y, dtable, B, D,
                     num_classes, accumulate);
  return check_launch("afd_embed_add_bwd");
}
int afd_pos_encoding(const int64_t* t, const float* inv_freq, float* temb, int B, int half, afd_stream_t st) {
  AFD_REQUIRE(t && inv_freq && temb && B > 0 && half > 0, "afd_pos_encoding: bad argument");
  hipLaunchKernelGGL(pos_encoding_k, dim3((B * half + 255) / 256), dim3(256), 0, as_stream(st), t, inv_freq, temb, B, half);
  return check_launch("afd_pos_encoding");
}
int afd_silu_linear_fwd(const float* temb, const float* w, const float* bias, float* out, int B, int K, int N, afd_stream_t st) {
  AFD_REQUIRE(temb && w && out && B > 0 && K > 0 && N > 0, "afd_silu_linear_fwd: bad argument");
  const long waves = (long)B * N;
  hipLaunchKernelGGL(silu_linear_fwd_k, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(st), temb, w, bias, out, B, K, N);
  return check_launch("afd_silu_linear_fwd");
}
int afd_silu_linear_fwd_batched(const float* temb, const afd_silu_desc* descs, int n, int B, int K, afd_stream_t st) {
  AFD_REQUIRE(temb && descs && n > 0 && n <= 8 && B > 0 && K > 0, "afd_silu_linear_fwd_batched: bad argument (1..8 layers)");
  SiluBatch d{};
  d.n = n; d.first[0] = 0;
  for (int i = 0; i < n; ++i) {
    AFD_REQUIRE(descs[i].w && descs[i].out && descs[i].N > 0, "afd_silu_linear_fwd_batched: bad descriptor %d", i);
    d.w[i] = descs[i].w; d.bias[i] = descs[i].bias; d.out[i] = descs[i].out; d.N[i] = descs[i].N;
    d.first[i + 1] = d.first[i] + descs[i].N;
  }
  const long waves = (long)B * ((d.first[n] + 7) / 8);
  hipLaunchKernelGGL(silu_linear_fwd_batched_k, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(st), temb, d, B, K);
  return check_launch("afd_silu_linear_fwd_batched");
}
int afd_gather_rows_batched(const int64_t* idx, const afd_silu_desc* descs, int n, int B, int rows, afd_stream_t st) {
  AFD_REQUIRE(idx && descs && n > 0 && n <= 8 && B > 0 && rows > 0, "afd_gather_rows_batched: bad argument (1..8 tables)");
  SiluBatch d{};
  d.n = n; d.first[0] = 0;
  for (int i = 0; i < n; ++i) {
    AFD_REQUIRE(descs[i].w && descs[i].out && descs[i].N > 0, "afd_gather_rows_batched: bad descriptor %d", i);
    d.w[i] = descs[i].w; d.bias[i] = nullptr; d.out[i] = descs[i].out; d.N[i] = descs[i].N;
    d.first[i + 1] = d.first[i] + descs[i].N;
  }
  const long total = (long)B * d.first[n];
  hipLaunchKernelGGL(gather_rows_batched_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(st), idx, d, B, rows);
  return check_launch("afd_gather_rows_batched");
}
int afd_silu_linear_bwd(const float* temb, const float* w, const float* dout, float* dw, float* dbias, float* dtemb,
                        int B, int K, int N, int accumulate, afd_stream_t st) {
  AFD_REQUIRE(temb && w && dout && dw && B > 0 && K > 0 && N > 0, "afd_silu_linear_bwd: bad argument");
  hipStream_t s = as_stream(st);
  AFD_REQUIRE(!dbias || (N + 31) / 32 <= (K + 31) / 32, "afd_silu_linear_bwd: N > K is not covered");   // (the bias row rides in the dW grid)
  hipLaunchKernelGGL(silu_linear_dw_k, dim3((K + 31) / 32, (N + 7) / 8 + (dbias ? 1 : 0)), dim3(256), 0, s, temb, dout, dw, dbias, B, K, N, accumulate);
  if (dtemb) hipLaunchKernelGGL(silu_linear_dx_k, dim3((unsigned)(((long)B * K + 255) / 256)), dim3(256), 0, s, temb, w, dout, dtemb, B, K, N);
  return check_launch("afd_silu_linear_bwd");
}

}  // extern "C"

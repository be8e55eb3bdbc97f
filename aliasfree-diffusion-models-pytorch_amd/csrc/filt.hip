// filt.hip -- F2/F3/F4: jinc*Kaiser filtered 2x resampling and the filtered GELU, gfx950.
//
// Two paths per op:
//   * fast path (N == 3, square power-of-two planes S in {4..64}): one lane owns one column of one
//     plane and marches down the rows; the whole column lives in registers, left/right neighbours
//     come from wave shuffles, so every input element is read from HBM exactly once and every
//     output written once (8 B/element forward, 12 B/element backward for the fused op).  A wave
//     carries 64/S planes side by side.  No LDS, no 2x-resolution intermediate in memory.
//   * general path (any N <= 15, any H, W): one thread per output element, direct tap loops;
//     the fused op composes up2 -> GELU -> down2 through a caller-provided workspace.
//
// Definitions restated from the reference (cross-correlation, zero 'same' padding with
// lo = (N-1)/2 on the low side, samples on even positions, no x4 gain): filtrs.py:71-94.
#include "common.h"

namespace afd {

// ------------------------------------------------------------------------------------------
// general path
// ------------------------------------------------------------------------------------------
__global__ void up2_fwd_gen(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W,
                            long xbs, long ybs, long total, Taps t, int N) {
  const int lo = (N - 1) / 2, H2 = 2 * H, W2 = 2 * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = i % W2, p = (i / W2) % H2;
    const long bc = i / ((long)W2 * H2);
    const int c = bc % C; const long b = bc / C;
    const float* xp = x + b * xbs + (long)c * H * W;
    float acc = 0.f;
    for (int a = 0; a < N; ++a) {
      const int zr = p + a - lo;
      if (zr < 0 || zr >= H2 || (zr & 1)) continue;
      for (int bb = 0; bb < N; ++bb) {
        const int zc = q + bb - lo;
        if (zc < 0 || zc >= W2 || (zc & 1)) continue;
        acc += t.k[a * N + bb] * xp[(zr >> 1) * W + (zc >> 1)];
      }
    }
    y[b * ybs + (long)c * H2 * W2 + (long)p * W2 + q] = acc;
  }
}

// adjoint of up2: dx[i,j] = sum_{a,b} k[a,b] * dU[2i - a + lo, 2j - b + lo]
__global__ void up2_bwd_gen(const float* __restrict__ dy, float* __restrict__ dx, int C, int H, int W,
                            long dybs, long dxbs, long total, Taps t, int N) {
  const int lo = (N - 1) / 2, H2 = 2 * H, W2 = 2 * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = i % W, r = (i / W) % H;
    const long bc = i / ((long)W * H);
    const int c = bc % C; const long b = bc / C;
    const float* dp = dy + b * dybs + (long)c * H2 * W2;
    float acc = 0.f;
    for (int a = 0; a < N; ++a) {
      const int p = 2 * r - a + lo;
      if (p < 0 || p >= H2) continue;
      for (int bb = 0; bb < N; ++bb) {
        const int q = 2 * j - bb + lo;
        if (q < 0 || q >= W2) continue;
        acc += t.k[a * N + bb] * dp[(long)p * W2 + q];
      }
    }
    dx[b * dxbs + (long)c * H * W + (long)r * W + j] = acc;
  }
}

__global__ void down2_fwd_gen(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W,
                              long xbs, long ybs, long total, Taps t, int N) {
  const int lo = (N - 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = i % Wo, r = (i / Wo) % Ho;
    const long bc = i / ((long)Wo * Ho);
    const int c = bc % C; const long b = bc / C;
    const float* xp = x + b * xbs + (long)c * H * W;
    float acc = 0.f;
    for (int a = 0; a < N; ++a) {
      const int p = 2 * r + a - lo;
      if (p < 0 || p >= H) continue;
      for (int bb = 0; bb < N; ++bb) {
        const int q = 2 * j + bb - lo;
        if (q < 0 || q >= W) continue;
        acc += t.k[a * N + bb] * xp[(long)p * W + q];
      }
    }
    y[b * ybs + (long)c * Ho * Wo + (long)r * Wo + j] = acc;
  }
}

// adjoint of down2: dX[p,q] = sum_{a,b : (p-a+lo) even, (q-b+lo) even} k[a,b] dD[(p-a+lo)/2, (q-b+lo)/2]
__global__ void down2_bwd_gen(const float* __restrict__ dy, float* __restrict__ dx, int C, int H, int W,
                              long dybs, long dxbs, long total, Taps t, int N) {
  const int lo = (N - 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = i % W, p = (i / W) % H;
    const long bc = i / ((long)W * H);
    const int c = bc % C; const long b = bc / C;
    const float* dp = dy + b * dybs + (long)c * Ho * Wo;
    float acc = 0.f;
    for (int a = 0; a < N; ++a) {
      const int rr = p - a + lo;
      if (rr < 0 || (rr & 1) || (rr >> 1) >= Ho) continue;
      for (int bb = 0; bb < N; ++bb) {
        const int cc = q - bb + lo;
        if (cc < 0 || (cc & 1) || (cc >> 1) >= Wo) continue;
        acc += t.k[a * N + bb] * dp[(long)(rr >> 1) * Wo + (cc >> 1)];
      }
    }
    dx[b * dxbs + (long)c * H * W + (long)p * W + q] = acc;
  }
}

// v = GroupNorm-apply(x) + res  (the optional prologue of F4), one element
__device__ __forceinline__ float prologue(const float* x, long idx, float sc, float sh, const float* res) {
  float v = x[idx] * sc + sh;
  if (res) v += res[idx];
  return v;
}

__device__ __forceinline__ void plane_affine(const float* stats, const float* gamma, const float* beta,
                                             long b, int c, float& sc, float& sh) {
  sc = 1.f; sh = 0.f;
  if (stats) {
    const float mean = stats[2 * b], rstd = stats[2 * b + 1];
    const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    sc = rstd * g;
    sh = be - mean * sc;
  }
}

// workspace helpers of the general fused path
__global__ void prologue_gen(const float* __restrict__ x, float* __restrict__ v, int C, long HW, long total,
                             const float* stats, const float* gamma, const float* beta, const float* res) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long bc = i / HW; const int c = bc % C; const long b = bc / C;
    float sc, sh; plane_affine(stats, gamma, beta, b, c, sc, sh);
    v[i] = prologue(x, i, sc, sh, res);
  }
}
__global__ void gelu_inplace_gen(float* u, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    u[i] = gelu_erf(u[i]);
}
__global__ void gelu_grad_mul_gen(const float* __restrict__ u, float* __restrict__ dg, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    dg[i] *= gelu_erf_grad(u[i]);
}

// ------------------------------------------------------------------------------------------
// exact-GELU by table: the fused kernels are bound by vector instructions (4 GELUs per element), and the rcp + exp
// form of erf costs ~19 of them.  GELU(u) = u Phi(u) and GELU'(u) = Phi(u) + u phi(u): Phi and GELU' are tabulated over
// SIGNED u on [-6, 6) as 192 cubic Hermite pieces of width 1/16 each (coefficients from fp64 erf / exp: interpolation
// error <= 2.2e-8 and 1.2e-7, below the 6e-7 of the A&S form it replaces; beyond +-6 both are 0 / 1 to 1e-9), copied to
// LDS by every workgroup: a lookup is fma + clamp + cvt + shift + fract, one ds_read_b128 (neighbouring pixels mostly
// hit the same piece: broadcast) and 3 fma -- 9 vector instructions per GELU, 8 per GELU' (the unsigned |u| tables of
// the first version needed a copysign and a recombination with u/2 on top: 11 / 10).
// The table is produced ON THE DEVICE by a one-thread-per-piece kernel at first use -- a launch, so it is legal under
// stream capture and allocates nothing (static device storage).
// ------------------------------------------------------------------------------------------
constexpr int kGeluPieces = 192;
constexpr float kGeluScale = 16.f, kGeluOffset = 96.f;
__device__ float4 g_gelu_tab[2 * kGeluPieces];        // [0, 192): Phi pieces; [192, 384): GELU' pieces

__global__ void gelu_table_init() {
  const int i = threadIdx.x;
  if (i >= kGeluPieces) return;
  const double hstep = 1.0 / 16.0, a0 = (i - 96) * hstep, a1 = a0 + hstep;
  const double isq2 = 0.70710678118654752440, isq2pi = 0.39894228040143267794;
  auto phi = [&](double a) { return isq2pi * exp(-0.5 * a * a); };
  auto Phi = [&](double a) { return 0.5 * erfc(-a * isq2); };
  auto gfun = [&](double a) { return Phi(a) + a * phi(a); };          // GELU'
  auto gder = [&](double a) { return (2.0 - a * a) * phi(a); };       // GELU''
  {
    const double f0 = Phi(a0), f1 = Phi(a1), d0 = hstep * phi(a0), d1 = hstep * phi(a1);
    g_gelu_tab[i] = make_float4((float)f0, (float)d0, (float)(3.0 * (f1 - f0) - 2.0 * d0 - d1), (float)(2.0 * (f0 - f1) + d0 + d1));
  }
  {
    const double f0 = gfun(a0), f1 = gfun(a1), d0 = hstep * gder(a0), d1 = hstep * gder(a1);
    g_gelu_tab[kGeluPieces + i] = make_float4((float)f0, (float)d0, (float)(3.0 * (f1 - f0) - 2.0 * d0 - d1), (float)(2.0 * (f0 - f1) + d0 + d1));
  }
}
static void ensure_gelu_table(hipStream_t s) {
  static bool done = false;
  if (done) return;
  hipLaunchKernelGGL(gelu_table_init, dim3(1), dim3(256), 0, s);
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  // a captured launch only runs at replay: keep launching (12 us once per call) until one eager launch has happened
  if (hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusNone) done = true;
}
// T[piece](t), piece = floor(16 u + 96) clamped to the table, t = frac
__device__ __forceinline__ float gelu_piece(float u, const float4* __restrict__ T) {
  const float sx = __builtin_amdgcn_fmed3f(fmaf(u, kGeluScale, kGeluOffset), 0.f, (float)kGeluPieces - 0.001f);
  const float4 c = T[(int)sx];
  const float t = __builtin_amdgcn_fractf(sx);
  return fmaf(fmaf(fmaf(c.w, t, c.z), t, c.y), t, c.x);
}
__device__ __forceinline__ float gelu_tab(float u, const float4* __restrict__ T) { return u * gelu_piece(u, T); }
__device__ __forceinline__ float gelu_grad_tab(float u, const float4* __restrict__ T) { return gelu_piece(u, T + kGeluPieces); }
__device__ __forceinline__ void load_gelu_table(float4* __restrict__ T, int first, int count) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) T[i] = g_gelu_tab[first + i];
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// fast path, N == 3, S x S planes.  lane = (plane slot, column)
// ------------------------------------------------------------------------------------------
template <int S>
struct Lane {
  static constexpr int PPW = kWave / S;                 // planes per wave
  int col; long plane; bool live;
  __device__ Lane(long planes) {
    const int lane = threadIdx.x & (kWave - 1);
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    col = lane % S;
    plane = wave * PPW + lane / S;
    live = plane < planes;
  }
  __device__ float left(float v) const { const float s = lane_left(v); return col == 0 ? 0.f : s; }
  __device__ float right(float v) const { const float s = lane_right(v); return col == S - 1 ? 0.f : s; }
  // the same in two halves, so that a pipelined kernel can request the shuffle in one stage and consume it in the next
  __device__ float left_mask(float raw) const { return col == 0 ? 0.f : raw; }
  __device__ float right_mask(float raw) const { return col == S - 1 ? 0.f : raw; }
};

// FULL (every fast-path kernel): the launch has no dead lanes (planes % planes-per-wave == 0 -- every production shape), so
// the per-row `if (live)` around loads and stores folds away.  With it each unrolled row is its own basic block (an
// exec-mask branch per store) and the compiler cannot move the next row's shuffles / table reads above the current
// row's arithmetic: every row then pays its three or four dependent LDS round trips in full (measured on the filtered
// GELU: 326 s_waitcnt in a 32-row kernel, vector pipe 64 % busy at 5 waves per SIMD).
// F2 fast: each lane writes its 2x2 polyphase block per input row as two float2 stores.
template <int S, bool FULL>
__global__ __launch_bounds__(256) void up2_fwd_n3(const float* __restrict__ x, float* __restrict__ y,
                                                  long planes, int C, long xbs, long ybs, Taps3 t) {
  Lane<S> L(planes);
  if (FULL && !L.live) return;            // a whole wave beyond the last plane (the grid is rounded up to 4 waves): wave-uniform
  const bool live = FULL || L.live;
  const long b = L.plane / C; const int c = L.plane % C;
  const float* xp = x + b * xbs + (long)c * S * S + L.col;
  float xv[S + 1];
#pragma unroll
  for (int i = 0; i < S; ++i) xv[i] = live ? xp[i * S] : 0.f;
  xv[S] = 0.f;
  float2* yp = reinterpret_cast<float2*>(y + b * ybs + (long)c * 4 * S * S) + L.col;
  float xr = L.right(xv[0]);
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const float xe = xv[i], xd = xv[i + 1];
    const float xdr = L.right(xd);
    float2 r0, r1;
    r0.x = t.k[4] * xe;                                             // U[2i,   2j]
    r0.y = t.k[3] * xe + t.k[5] * xr;                               // U[2i,   2j+1]
    r1.x = t.k[1] * xe + t.k[7] * xd;                               // U[2i+1, 2j]
    r1.y = t.k[0] * xe + t.k[2] * xr + t.k[6] * xd + t.k[8] * xdr;  // U[2i+1, 2j+1]
    if (live) { yp[(2 * i) * S] = r0; yp[(2 * i + 1) * S] = r1; }
    xr = xdr;
  }
}

// F3 fast: input plane is 2S x 2S, lane j owns input columns 2j, 2j+1 (float2 loads).
template <int S, bool FULL>   // S = OUTPUT side
__global__ __launch_bounds__(256) void down2_fwd_n3(const float* __restrict__ x, float* __restrict__ y,
                                                    long planes, int C, long xbs, long ybs, Taps3 t) {
  Lane<S> L(planes);
  if (FULL && !L.live) return;            // a whole wave beyond the last plane (the grid is rounded up to 4 waves): wave-uniform
  const bool live = FULL || L.live;
  const long b = L.plane / C; const int c = L.plane % C;
  const float2* xp = reinterpret_cast<const float2*>(x + b * xbs + (long)c * 4 * S * S) + L.col;
  float* yp = y + b * ybs + (long)c * S * S + L.col;
  float2 prev = make_float2(0.f, 0.f);     // row 2i-1 (zero above the image)
  float prev_l = 0.f;
  float2 rows[2 * S];
#pragma unroll
  for (int r = 0; r < 2 * S; ++r) rows[r] = live ? xp[r * S] : make_float2(0.f, 0.f);
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const float2 mid = rows[2 * i], nxt = rows[2 * i + 1];
    const float mid_l = L.left(mid.y), nxt_l = L.left(nxt.y);
    const float d = t.k[0] * prev_l + t.k[1] * prev.x + t.k[2] * prev.y
                  + t.k[3] * mid_l  + t.k[4] * mid.x  + t.k[5] * mid.y
                  + t.k[6] * nxt_l  + t.k[7] * nxt.x  + t.k[8] * nxt.y;
    if (live) yp[i * S] = d;
    prev = nxt; prev_l = nxt_l;
  }
}

// ---- the filtered GELU as a three-stage software pipeline over the rows of a column ---------------------------------
// A row's work is a chain through THREE dependent LDS round trips: the right-neighbour shuffle of x (for U11), the four
// table reads of the GELU pieces, the left-neighbour shuffles of g01 / g11 (for the 3x3 down filter).  Issued in plain
// program order every row waits for each of them in turn (~4 x 100+ cycles against ~270 cycles of arithmetic: the vector
// pipe was 64 % busy at 5 waves per SIMD).  Here row k's table reads, row k-1's interpolation + shuffles and row k-2's
// output taps are interleaved in one unrolled stream, so every LDS result is consumed one full row of arithmetic after
// it was requested (counted s_waitcnt: younger requests stay in flight).
struct ActRow { float U[4], t[4]; float4 c[4]; };            // stage 1: filter outputs on the 2x grid, table pieces requested
struct ActG { float g00, g01, g10, g11, g01l, g11l; };       // stage 2: GELU values + the left neighbours' odd columns

__device__ __forceinline__ void act_lookup(float U, const float4* __restrict__ T, float& t, float4& c) {
  const float sx = __builtin_amdgcn_fmed3f(fmaf(U, kGeluScale, kGeluOffset), 0.f, (float)kGeluPieces - 0.001f);
  c = T[(int)sx];
  t = __builtin_amdgcn_fractf(sx);
}
__device__ __forceinline__ float act_cubic(const float4& c, float t) { return fmaf(fmaf(fmaf(c.w, t, c.z), t, c.y), t, c.x); }

// F4 fast forward: y = down2(gelu(up2(v))), v = prologue(x).
// STATS (small samples, C * S <= 1024): the workgroup IS one sample (blockDim = C * S threads, one lane per column of every
// plane), so it computes the GroupNorm(1, C) statistics itself -- two-pass, from the columns it already holds in registers --
// applies them and writes {mean, rstd} to stats_out for the backward kernels: no separate statistics launch.
template <int S, bool FULL, bool STATS = false>
__global__ __launch_bounds__(STATS ? 1024 : 256) void filt_act_fwd_n3(const float* __restrict__ x, float* __restrict__ y,
                                                       long planes, int C, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ res, Taps3 u, Taps3 d,
                                                       float* __restrict__ stats_out = nullptr, float eps = 0.f) {
  __shared__ float4 T[kGeluPieces];
  load_gelu_table(T, 0, kGeluPieces);
  Lane<S> L(planes);
  const long b = L.plane / C; const int c = L.plane % C;
  const long base = L.plane * (long)S * S + L.col;
  if (FULL && !L.live) return;            // a whole wave beyond the last plane (the grid is rounded up to 4 waves): wave-uniform
  const bool live = FULL || L.live;
  float sc, sh;
  float xv[S + 2];
  if (STATS) {
    __shared__ float red[16];
#pragma unroll
    for (int i = 0; i < S; ++i) xv[i] = x[base + i * S];
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < S; ++i) sm += xv[i];
    const float n = (float)C * S * S;
    const float mean = block_sum(sm, red) / n;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < S; ++i) { const float dd = xv[i] - mean; m2 = fmaf(dd, dd, m2); }
    const float rstd = 1.0f / sqrtf(block_sum(m2, red) / n + eps);
    if (threadIdx.x == 0) { stats_out[2 * b] = mean; stats_out[2 * b + 1] = rstd; }
    sc = rstd * (gamma ? gamma[c] : 1.f);
    sh = (beta ? beta[c] : 0.f) - mean * sc;
#pragma unroll
    for (int i = 0; i < S; ++i) xv[i] = fmaf(xv[i], sc, sh);
  } else {
    if (live) plane_affine(stats, gamma, beta, b, c, sc, sh); else { sc = 1.f; sh = 0.f; }
#pragma unroll
    for (int i = 0; i < S; ++i) xv[i] = live ? fmaf(x[base + i * S], sc, sh) : 0.f;
  }
  if (res) {                                       // ONE uniform branch around all residual loads (not one per row)
#pragma unroll
    for (int i = 0; i < S; ++i) xv[i] += live ? res[base + i * S] : 0.f;
  }
  xv[S] = 0.f; xv[S + 1] = 0.f;
  ActRow row[2];
  ActG gg[2];
  float g10p = 0.f, g11p = 0.f, g11lp = 0.f;       // odd row of the previous input row (G[2i-1, .])
  float xr0 = L.right(xv[0]), xr1 = lane_right(xv[1]);    // right neighbours of rows k, k + 1 (rolling; xr1 still unmasked)
#pragma unroll
  for (int k = 0; k < S + 2; ++k) {
    if (k < S) {                                   // stage 1 of row k
      ActRow& r = row[k & 1];
      const float xe = xv[k], xd = xv[k + 1];
      xr1 = L.right_mask(xr1);                     // (requested one row ago)
      r.U[0] = u.k[4] * xe;
      r.U[1] = u.k[3] * xe + u.k[5] * xr0;
      r.U[2] = u.k[1] * xe + u.k[7] * xd;
      r.U[3] = u.k[0] * xe + u.k[2] * xr0 + u.k[6] * xd + u.k[8] * xr1;
#pragma unroll
      for (int q = 0; q < 4; ++q) act_lookup(r.U[q], T, r.t[q], r.c[q]);
      xr0 = xr1;
      xr1 = (k + 2 <= S) ? lane_right(xv[k + 2]) : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);             // keep the stages in this order: the scheduler would otherwise pull each
    //                                                consumer back up to its request to save registers, and re-serialise
    if (k >= 2) {                                  // stage 3 of row k - 2
      const ActG& g = gg[k & 1];
      const float g01l = L.left_mask(g.g01l), g11l = L.left_mask(g.g11l);      // (requested one row ago)
      const float out = d.k[0] * g11lp + d.k[1] * g10p + d.k[2] * g11p
                      + d.k[3] * g01l + d.k[4] * g.g00 + d.k[5] * g.g01
                      + d.k[6] * g11l + d.k[7] * g.g10 + d.k[8] * g.g11;
      if (live) y[base + (k - 2) * S] = out;
      g10p = g.g10; g11p = g.g11; g11lp = g11l;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (k >= 1 && k <= S) {                        // stage 2 of row k - 1
      const ActRow& r = row[(k - 1) & 1];
      ActG& g = gg[(k - 1) & 1];
      g.g00 = r.U[0] * act_cubic(r.c[0], r.t[0]);
      g.g01 = r.U[1] * act_cubic(r.c[1], r.t[1]);
      g.g10 = r.U[2] * act_cubic(r.c[2], r.t[2]);
      g.g11 = r.U[3] * act_cubic(r.c[3], r.t[3]);
      g.g01l = lane_left(g.g01);                   // unmasked: stage 3 applies the plane border
      g.g11l = lane_left(g.g11);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// F4 fast backward: dv from (x, dy); U is recomputed, nothing at 2x resolution touches memory.
//   dG[2i,2j]     = d11 dD[i,j]
//   dG[2i,2j+1]   = d12 dD[i,j] + d10 dD[i,j+1]
//   dG[2i+1,2j]   = d21 dD[i,j] + d01 dD[i+1,j]
//   dG[2i+1,2j+1] = d22 dD[i,j] + d20 dD[i,j+1] + d02 dD[i+1,j] + d00 dD[i+1,j+1]
//   dU = dG * gelu'(U);   dv[i,j] = sum_{a,b} u[a,b] dU[2i-a+1, 2j-b+1]
// Same three-stage pipeline as the forward: stage 1 = U, dG and the GELU' table requests of row k, stage 2 = dU and its
// left-neighbour shuffles of row k - 1, stage 3 = the up filter's adjoint taps of row k - 2.
struct ActRowB { float t[4], dG[4]; float4 c[4]; };
struct ActGB { float dU00, dU01, dU10, dU11, dU01l, dU11l; };

// GNB (small samples, the workgroup IS one sample: blockDim = C * S, see the forward): the kernel also finishes GroupNorm's
// backward -- dv stays in registers, the sample's two sums meet in LDS, dx = rstd (gamma dv - m1 - xhat m2) leaves instead of
// dv (dv itself only when the residual branch needs it: dres = dv) -- so those sites have no gn_bwd_apply launch and no dv
// round trip.
template <int S, bool FULL, bool GNB = false>
__global__ __launch_bounds__(GNB ? 1024 : 256) void filt_act_bwd_n3(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ dv, long planes, int C,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ res,
                                                       float* __restrict__ part, Taps3 u, Taps3 d, float* __restrict__ dx = nullptr) {
  __shared__ float4 Tb[kGeluPieces];                 // the GELU' pieces only
  load_gelu_table(Tb, kGeluPieces, kGeluPieces);
  Lane<S> L(planes);
  const long b = L.plane / C; const int c = L.plane % C;
  const long base = L.plane * (long)S * S + L.col;
  if (FULL && !L.live) return;            // a whole wave beyond the last plane (the grid is rounded up to 4 waves): wave-uniform
  const bool live = FULL || L.live;
  float sc, sh;
  if (live) plane_affine(stats, gamma, beta, b, c, sc, sh); else { sc = 1.f; sh = 0.f; }
  float mean = 0.f, rstd = 1.f;
  if (part && live) { mean = stats[2 * b]; rstd = stats[2 * b + 1]; }
  float xv[S + 2], gv[S + 2];
  float a1 = 0.f, a2 = 0.f;                         // GroupNorm-backward plane sums: sum dv*xhat, sum dv
  float hx[S];                                      // xhat of the raw input (only needed for the partials)
  float dvc[GNB ? S : 1];                           // GNB: this lane's column of dv
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const float h = live ? x[base + i * S] : 0.f;
    xv[i] = live ? fmaf(h, sc, sh) : 0.f;
    hx[i] = (h - mean) * rstd;
    gv[i] = live ? dy[base + i * S] : 0.f;
  }
  if (res) {                                       // ONE uniform branch around all residual loads
#pragma unroll
    for (int i = 0; i < S; ++i) xv[i] += live ? res[base + i * S] : 0.f;
  }
  xv[S] = 0.f; gv[S] = 0.f; xv[S + 1] = 0.f; gv[S + 1] = 0.f;
  ActRowB row[2];
  ActGB gg[2];
  float u10p = 0.f, u11p = 0.f, u11lp = 0.f;       // dU on the odd row of the previous input row
  float xr0 = L.right(xv[0]), xr1 = lane_right(xv[1]), gr0 = L.right(gv[0]), gr1 = lane_right(gv[1]);   // (xr1, gr1 unmasked)
#pragma unroll
  for (int k = 0; k < S + 2; ++k) {
    if (k < S) {                                   // stage 1 of row k
      ActRowB& r = row[k & 1];
      const float xe = xv[k], xd = xv[k + 1], ge = gv[k], gd = gv[k + 1];
      xr1 = L.right_mask(xr1); gr1 = L.right_mask(gr1);                       // (requested one row ago)
      const float U00 = u.k[4] * xe;
      const float U01 = u.k[3] * xe + u.k[5] * xr0;
      const float U10 = u.k[1] * xe + u.k[7] * xd;
      const float U11 = u.k[0] * xe + u.k[2] * xr0 + u.k[6] * xd + u.k[8] * xr1;
      act_lookup(U00, Tb, r.t[0], r.c[0]);
      act_lookup(U01, Tb, r.t[1], r.c[1]);
      act_lookup(U10, Tb, r.t[2], r.c[2]);
      act_lookup(U11, Tb, r.t[3], r.c[3]);
      r.dG[0] = d.k[4] * ge;
      r.dG[1] = d.k[5] * ge + d.k[3] * gr0;
      r.dG[2] = d.k[7] * ge + d.k[1] * gd;
      r.dG[3] = d.k[8] * ge + d.k[6] * gr0 + d.k[2] * gd + d.k[0] * gr1;
      xr0 = xr1; gr0 = gr1;
      xr1 = (k + 2 <= S) ? lane_right(xv[k + 2]) : 0.f;
      gr1 = (k + 2 <= S) ? lane_right(gv[k + 2]) : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (k >= 2) {                                  // stage 3 of row k - 2
      const ActGB& g = gg[k & 1];
      const float dU01l = L.left_mask(g.dU01l), dU11l = L.left_mask(g.dU11l);   // (requested one row ago)
      const float out = u.k[0] * g.dU11 + u.k[1] * g.dU10 + u.k[2] * dU11l
                      + u.k[3] * g.dU01 + u.k[4] * g.dU00 + u.k[5] * dU01l
                      + u.k[6] * u11p + u.k[7] * u10p + u.k[8] * u11lp;
      if (GNB) { dvc[k - 2] = out; if (dv) dv[base + (k - 2) * S] = out; }
      else if (live) dv[base + (k - 2) * S] = out;
      a1 += out * hx[k - 2]; a2 += out;
      u10p = g.dU10; u11p = g.dU11; u11lp = dU11l;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (k >= 1 && k <= S) {                        // stage 2 of row k - 1
      const ActRowB& r = row[(k - 1) & 1];
      ActGB& g = gg[(k - 1) & 1];
      g.dU00 = r.dG[0] * act_cubic(r.c[0], r.t[0]);
      g.dU01 = r.dG[1] * act_cubic(r.c[1], r.t[1]);
      g.dU10 = r.dG[2] * act_cubic(r.c[2], r.t[2]);
      g.dU11 = r.dG[3] * act_cubic(r.c[3], r.t[3]);
      g.dU01l = lane_left(g.dU01);                 // unmasked: stage 3 applies the plane border
      g.dU11l = lane_left(g.dU11);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (GNB) {                                        // the sample's sums (every lane of the workgroup belongs to sample b)
    __shared__ float red[16];
    const float g = gamma ? gamma[c] : 1.f, n = (float)C * S * S;
    const float m1 = block_sum(g * a2, red) / n;    // mean(gamma dv)
    const float m2 = block_sum(g * a1, red) / n;    // mean(gamma dv xhat)
#pragma unroll
    for (int i = 0; i < S; ++i) dx[base + i * S] = rstd * (g * dvc[i] - m1 - hx[i] * m2);
  }
  if (part) {                                       // reduce over the S lanes that share the plane (fixed butterfly)
#pragma unroll
    for (int o = S / 2; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o, kWave); a2 += __shfl_xor(a2, o, kWave); }
    if (live && L.col == 0) { part[(2 * b) * C + c] = a1; part[(2 * b + 1) * C + c] = a2; }
  }
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------
static inline int gs_grid(long total, int block = 256) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}
static inline bool fast_side(int H, int W) { return H == W && (H == 4 || H == 8 || H == 16 || H == 32 || H == 64); }
static inline Taps load_taps(const float* k, int N) { Taps t; for (int i = 0; i < N * N; ++i) t.k[i] = k[i]; return t; }
static inline Taps3 load_taps3(const float* k, bool rot180 = false) {
  Taps3 t; for (int i = 0; i < 9; ++i) t.k[i] = k[rot180 ? 8 - i : i]; return t;
}


// (`planes` must be in scope: FULL = no dead lanes in the launch)
#define AFD_DISPATCH_SF(S_, KERNEL, GRID_ARGS, ...)                                                      \
  if (planes % (64 / (S_)) == 0) hipLaunchKernelGGL((KERNEL<S_, true>), GRID_ARGS, __VA_ARGS__);          \
  else hipLaunchKernelGGL((KERNEL<S_, false>), GRID_ARGS, __VA_ARGS__)
#define AFD_DISPATCH_S(S_, KERNEL, GRID_ARGS, ...)                                          \
  switch (S_) {                                                                              \
    case 4:  AFD_DISPATCH_SF(4, KERNEL, GRID_ARGS, __VA_ARGS__); break;                      \
    case 8:  AFD_DISPATCH_SF(8, KERNEL, GRID_ARGS, __VA_ARGS__); break;                      \
    case 16: AFD_DISPATCH_SF(16, KERNEL, GRID_ARGS, __VA_ARGS__); break;                     \
    case 32: AFD_DISPATCH_SF(32, KERNEL, GRID_ARGS, __VA_ARGS__); break;                     \
    default: AFD_DISPATCH_SF(64, KERNEL, GRID_ARGS, __VA_ARGS__); break;                     \
  }

static inline dim3 fast_grid(long planes, int S) {
  const long ppw = 64 / S, waves = (planes + ppw - 1) / ppw;
  return dim3((unsigned)((waves + 3) / 4));
}

static int check_common(const char* fn, const void* a, const void* b, int B, int C, int H, int W, const float* taps, int N) {
  AFD_REQUIRE(a && b && taps, "%s: NULL pointer", fn);
  AFD_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "%s: non-positive shape B=%d C=%d H=%d W=%d", fn, B, C, H, W);
  AFD_REQUIRE(N >= 1 && N <= AFD_MAX_TAPS, "%s: N=%d outside [1,%d]", fn, N, AFD_MAX_TAPS);
  return AFD_OK;
}

static int up2_like(const char* fn, const float* x, float* y, int B, int C, int H, int W, long xbs, long ybs,
                    const float* taps, int N, bool rot, hipStream_t s) {
  // y (2H x 2W) = up2(x; taps)   [rot: taps rotated by 180 deg -- used as the adjoint of down2 for N == 3]
  if (!xbs) xbs = (long)C * H * W;
  if (!ybs) ybs = (long)C * 4 * H * W;
  const long planes = (long)B * C;
  if (N == 3 && fast_side(H, W)) {
    const Taps3 t = load_taps3(taps, rot);
    AFD_DISPATCH_S(H, up2_fwd_n3, fast_grid(planes, H), dim3(256), 0, s, x, y, planes, C, xbs, ybs, t);
  } else {
    AFD_REQUIRE(!rot, "%s: internal: rot180 only on the fast path", fn);
    const long total = planes * 4 * H * W;
    hipLaunchKernelGGL(up2_fwd_gen, dim3(gs_grid(total)), dim3(256), 0, s, x, y, C, H, W, xbs, ybs, total, load_taps(taps, N), N);
  }
  return check_launch(fn);
}

static int down2_like(const char* fn, const float* x, float* y, int B, int C, int H, int W, long xbs, long ybs,
                      const float* taps, int N, bool rot, hipStream_t s) {
  // y (ceil(H/2) x ceil(W/2)) = down2(x (H x W); taps)
  if (!xbs) xbs = (long)C * H * W;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  if (!ybs) ybs = (long)C * Ho * Wo;
  const long planes = (long)B * C;
  if (N == 3 && (H % 2 == 0) && (W % 2 == 0) && fast_side(Ho, Wo)) {
    const Taps3 t = load_taps3(taps, rot);
    AFD_DISPATCH_S(Ho, down2_fwd_n3, fast_grid(planes, Ho), dim3(256), 0, s, x, y, planes, C, xbs, ybs, t);
  } else {
    AFD_REQUIRE(!rot, "%s: internal: rot180 only on the fast path", fn);
    const long total = planes * Ho * Wo;
    hipLaunchKernelGGL(down2_fwd_gen, dim3(gs_grid(total)), dim3(256), 0, s, x, y, C, H, W, xbs, ybs, total, load_taps(taps, N), N);
  }
  return check_launch(fn);
}

}  // namespace afd

using namespace afd;

extern "C" {

int afd_filt_up2_fwd(const float* x, float* y, int B, int C, int H, int W, long xbs, long ybs,
                     const float* taps, int N, afd_stream_t stream) {
  if (int e = check_common("afd_filt_up2_fwd", x, y, B, C, H, W, taps, N)) return e;
  return up2_like("afd_filt_up2_fwd", x, y, B, C, H, W, xbs, ybs, taps, N, false, as_stream(stream));
}

int afd_filt_up2_bwd(const float* dy, float* dx, int B, int C, int H, int W, long dybs, long dxbs,
                     const float* taps, int N, afd_stream_t stream) {
  if (int e = check_common("afd_filt_up2_bwd", dy, dx, B, C, H, W, taps, N)) return e;
  hipStream_t s = as_stream(stream);
  if (N == 3 && fast_side(H, W))   // adjoint of up2 == down2 with the taps rotated by 180 degrees
    return down2_like("afd_filt_up2_bwd", dy, dx, B, C, 2 * H, 2 * W, dybs, dxbs, taps, N, true, s);
  if (!dybs) dybs = (long)C * 4 * H * W;
  if (!dxbs) dxbs = (long)C * H * W;
  const long total = (long)B * C * H * W;
  hipLaunchKernelGGL(up2_bwd_gen, dim3(gs_grid(total)), dim3(256), 0, s, dy, dx, C, H, W, dybs, dxbs, total, load_taps(taps, N), N);
  return check_launch("afd_filt_up2_bwd");
}

int afd_filt_down2_fwd(const float* x, float* y, int B, int C, int H, int W, long xbs, long ybs,
                       const float* taps, int N, afd_stream_t stream) {
  if (int e = check_common("afd_filt_down2_fwd", x, y, B, C, H, W, taps, N)) return e;
  return down2_like("afd_filt_down2_fwd", x, y, B, C, H, W, xbs, ybs, taps, N, false, as_stream(stream));
}

int afd_filt_down2_bwd(const float* dy, float* dx, int B, int C, int H, int W, long dybs, long dxbs,
                       const float* taps, int N, afd_stream_t stream) {
  if (int e = check_common("afd_filt_down2_bwd", dy, dx, B, C, H, W, taps, N)) return e;
  hipStream_t s = as_stream(stream);
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  if (N == 3 && (H % 2 == 0) && (W % 2 == 0) && fast_side(Ho, Wo))   // adjoint of down2 == up2 with rotated taps
    return up2_like("afd_filt_down2_bwd", dy, dx, B, C, Ho, Wo, dybs, dxbs, taps, N, true, s);
  if (!dybs) dybs = (long)C * Ho * Wo;
  if (!dxbs) dxbs = (long)C * H * W;
  const long total = (long)B * C * H * W;
  hipLaunchKernelGGL(down2_bwd_gen, dim3(gs_grid(total)), dim3(256), 0, s, dy, dx, C, H, W, dybs, dxbs, total, load_taps(taps, N), N);
  return check_launch("afd_filt_down2_bwd");
}

size_t afd_filt_act_workspace_bytes(int B, int C, int H, int W, int N, int backward) {
  if (N == 3 && fast_side(H, W)) return 0;
  const size_t e = (size_t)B * C * H * W;
  // fwd: v (e) + U (4e);  bwd: v (e) + U (4e) + dG (4e)
  return sizeof(float) * (backward ? 9 * e : 5 * e);
}

int afd_filt_act_fwd(const float* x, float* y, int B, int C, int H, int W,
                     const float* stats, const float* gamma, const float* beta, const float* res,
                     const float* taps_up, const float* taps_down, int N, void* workspace, afd_stream_t stream) {
  if (int e = check_common("afd_filt_act_fwd", x, y, B, C, H, W, taps_up, N)) return e;
  AFD_REQUIRE(taps_down, "afd_filt_act_fwd: taps_down is NULL");
  hipStream_t s = as_stream(stream);
  const long planes = (long)B * C;
  if (N == 3 && fast_side(H, W)) {
    const Taps3 u = load_taps3(taps_up), d = load_taps3(taps_down);
    ensure_gelu_table(s);
    AFD_DISPATCH_S(H, filt_act_fwd_n3, fast_grid(planes, H), dim3(256), 0, s, x, y, planes, C, stats, gamma, beta, res, u, d);
    return check_launch("afd_filt_act_fwd");
  }
  AFD_REQUIRE(workspace, "afd_filt_act_fwd: this shape (N=%d, %dx%d) needs a workspace", N, H, W);
  const long e = planes * H * W;
  float* v = static_cast<float*>(workspace);
  float* U = v + e;
  hipLaunchKernelGGL(prologue_gen, dim3(gs_grid(e)), dim3(256), 0, s, x, v, C, (long)H * W, e, stats, gamma, beta, res);
  hipLaunchKernelGGL(up2_fwd_gen, dim3(gs_grid(4 * e)), dim3(256), 0, s, v, U, C, H, W, (long)C * H * W, (long)C * 4 * H * W, 4 * e, load_taps(taps_up, N), N);
  hipLaunchKernelGGL(gelu_inplace_gen, dim3(gs_grid(4 * e)), dim3(256), 0, s, U, 4 * e);
  hipLaunchKernelGGL(down2_fwd_gen, dim3(gs_grid(e)), dim3(256), 0, s, U, y, C, 2 * H, 2 * W, (long)C * 4 * H * W, (long)C * H * W, e, load_taps(taps_down, N), N);
  return check_launch("afd_filt_act_fwd");
}

size_t afd_filt_act_fwd_gn_supported(int C, int H, int W, int N) {
  return (N == 3 && H == W && (H == 4 || H == 8 || H == 16) && (long)C * H <= 1024 && ((long)C * H) % 64 == 0) ? 1 : 0;
}

int afd_filt_act_fwd_gn(const float* x, float* y, int B, int C, int H, int W, float eps, float* stats_out,
                        const float* gamma, const float* beta, const float* res,
                        const float* taps_up, const float* taps_down, int N, afd_stream_t stream) {
  if (int e = check_common("afd_filt_act_fwd_gn", x, y, B, C, H, W, taps_up, N)) return e;
  AFD_REQUIRE(taps_down && stats_out, "afd_filt_act_fwd_gn: NULL pointer");
  AFD_REQUIRE(afd_filt_act_fwd_gn_supported(C, H, W, N), "afd_filt_act_fwd_gn: shape (C=%d, %dx%d, N=%d) is not covered", C, H, W, N);
  hipStream_t s = as_stream(stream);
  const Taps3 u = load_taps3(taps_up), d = load_taps3(taps_down);
  ensure_gelu_table(s);
  const long planes = (long)B * C;
  const dim3 grid((unsigned)B), block((unsigned)(C * H));                    // one workgroup = one sample: a lane per column of every plane
  if (H == 4) hipLaunchKernelGGL((filt_act_fwd_n3<4, true, true>), grid, block, 0, s, x, y, planes, C, nullptr, gamma, beta, res, u, d, stats_out, eps);
  else if (H == 8) hipLaunchKernelGGL((filt_act_fwd_n3<8, true, true>), grid, block, 0, s, x, y, planes, C, nullptr, gamma, beta, res, u, d, stats_out, eps);
  else hipLaunchKernelGGL((filt_act_fwd_n3<16, true, true>), grid, block, 0, s, x, y, planes, C, nullptr, gamma, beta, res, u, d, stats_out, eps);
  return check_launch("afd_filt_act_fwd_gn");
}

int afd_filt_act_bwd_gn(const float* x, const float* dy, float* dx, float* dres, int B, int C, int H, int W, const float* stats,
                        const float* gamma, const float* beta, const float* res, const float* taps_up, const float* taps_down, int N,
                        float* gn_partials, afd_stream_t stream) {
  if (int e = check_common("afd_filt_act_bwd_gn", x, dy, B, C, H, W, taps_up, N)) return e;
  AFD_REQUIRE(dx && taps_down && stats && gn_partials, "afd_filt_act_bwd_gn: NULL pointer");
  AFD_REQUIRE(afd_filt_act_fwd_gn_supported(C, H, W, N), "afd_filt_act_bwd_gn: shape (C=%d, %dx%d, N=%d) is not covered", C, H, W, N);
  hipStream_t s = as_stream(stream);
  const Taps3 u = load_taps3(taps_up), d = load_taps3(taps_down);
  ensure_gelu_table(s);
  const long planes = (long)B * C;
  const dim3 grid((unsigned)B), block((unsigned)(C * H));
  if (H == 4) hipLaunchKernelGGL((filt_act_bwd_n3<4, true, true>), grid, block, 0, s, x, dy, dres, planes, C, stats, gamma, beta, res, gn_partials, u, d, dx);
  else if (H == 8) hipLaunchKernelGGL((filt_act_bwd_n3<8, true, true>), grid, block, 0, s, x, dy, dres, planes, C, stats, gamma, beta, res, gn_partials, u, d, dx);
  else hipLaunchKernelGGL((filt_act_bwd_n3<16, true, true>), grid, block, 0, s, x, dy, dres, planes, C, stats, gamma, beta, res, gn_partials, u, d, dx);
  return check_launch("afd_filt_act_bwd_gn");
}

int afd_filt_act_bwd(const float* x, const float* dy, float* dv, int B, int C, int H, int W,
                     const float* stats, const float* gamma, const float* beta, const float* res,
                     const float* taps_up, const float* taps_down, int N, void* workspace, float* gn_partials,
                     afd_stream_t stream) {
  if (int e = check_common("afd_filt_act_bwd", x, dy, B, C, H, W, taps_up, N)) return e;
  AFD_REQUIRE(dv && taps_down, "afd_filt_act_bwd: NULL pointer");
  hipStream_t s = as_stream(stream);
  const long planes = (long)B * C;
  if (N == 3 && fast_side(H, W)) {
    const Taps3 u = load_taps3(taps_up), d = load_taps3(taps_down);
    AFD_REQUIRE(!gn_partials || stats, "afd_filt_act_bwd: gn_partials needs stats");
    ensure_gelu_table(s);
    AFD_DISPATCH_S(H, filt_act_bwd_n3, fast_grid(planes, H), dim3(256), 0, s, x, dy, dv, planes, C, stats, gamma, beta, res, gn_partials, u, d);
    return check_launch("afd_filt_act_bwd");
  }
  AFD_REQUIRE(workspace, "afd_filt_act_bwd: this shape (N=%d, %dx%d) needs a workspace", N, H, W);
  AFD_REQUIRE(!gn_partials, "afd_filt_act_bwd: gn_partials are produced on the fused fast path only (workspace_bytes == 0)");
  const long e = planes * H * W;
  float* v = static_cast<float*>(workspace);
  float* U = v + e;
  float* dG = U + 4 * e;
  const long cs = (long)C * H * W, cs4 = 4 * cs;
  hipLaunchKernelGGL(prologue_gen, dim3(gs_grid(e)), dim3(256), 0, s, x, v, C, (long)H * W, e, stats, gamma, beta, res);
  hipLaunchKernelGGL(up2_fwd_gen, dim3(gs_grid(4 * e)), dim3(256), 0, s, v, U, C, H, W, cs, cs4, 4 * e, load_taps(taps_up, N), N);
  hipLaunchKernelGGL(down2_bwd_gen, dim3(gs_grid(4 * e)), dim3(256), 0, s, dy, dG, C, 2 * H, 2 * W, cs, cs4, 4 * e, load_taps(taps_down, N), N);
  hipLaunchKernelGGL(gelu_grad_mul_gen, dim3(gs_grid(4 * e)), dim3(256), 0, s, U, dG, 4 * e);
  hipLaunchKernelGGL(up2_bwd_gen, dim3(gs_grid(e)), dim3(256), 0, s, dG, dv, C, H, W, cs4, cs, e, load_taps(taps_up, N), N);
  return check_launch("afd_filt_act_bwd");
}

}  // extern "C"

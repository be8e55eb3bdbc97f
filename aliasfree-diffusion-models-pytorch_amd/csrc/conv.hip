// conv.hip -- F5/F10: 3x3 (pad 1) and 1x1 convolution on NCHW fp32 as implicit GEMM on the
// exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32), plus a direct path for tiny channel counts.
//
// Orientation (all three kernels): the MFMA "A" operand is the weight / dY side (rows = output
// channels), the "B" operand is the activation side (columns = pixels or input channels), so that
//   forward / dgrad:  D[n][pixel]   -> 32 consecutive pixels per accumulator register: 128-B stores
//   wgrad:            D[cout][cin]  per tap, reduction over pixels (split-K over pixel chunks,
//                     partial slabs + deterministic reduce; no atomics)
// Activations stay NCHW: a pixel tile is 128 (64 for wgrad) consecutive global pixels = whole rows
// of one image or whole small images, staged into LDS with a zero halo; weights are staged straight
// from OIHW (no transform pass) into rows of odd stride, which makes every fragment read
// conflict-free.  fp32 in / fp32 accumulate: bitwise an fma chain, no reduced precision anywhere.
#include <cstdlib>
#include <algorithm>
#include "common.h"

namespace afd {

using f32x16 = __attribute__((ext_vector_type(16))) float;

static inline int gs_grid(long total, int block = 256) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

// ------------------------------------------------------------------------------------------
// direct kernels (any shape; used for Cin/Cout < 8 and for planes the tiled path cannot cut)
// ------------------------------------------------------------------------------------------
template <int KS>
__global__ void conv_direct_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                const float* __restrict__ res, float* __restrict__ y,
                                int Cin, int Cout, int H, int W, int act, long total) {
  constexpr int T = KS * KS, R = KS / 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = i % W, yy = (i / W) % H;
    const long bc = i / ((long)W * H);
    const int co = bc % Cout; const long b = bc / Cout;
    float acc = bias ? bias[co] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
      const float* xp = x + (b * Cin + ci) * (long)H * W;
      const float* wp = w + ((long)co * Cin + ci) * T;
#pragma unroll
      for (int a = 0; a < KS; ++a) {
        const int r = yy + a - R;
        if (r < 0 || r >= H) continue;
#pragma unroll
        for (int c = 0; c < KS; ++c) {
          const int q = xx + c - R;
          if (q < 0 || q >= W) continue;
          acc += wp[a * KS + c] * xp[(long)r * W + q];
        }
      }
    }
    if (act == 1) acc = gelu_erf(acc);
    if (res) acc += res[i];
    y[i] = acc;
  }
}

template <int KS>
__global__ void conv_direct_dgrad(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                  int Cin, int Cout, int H, int W, long total) {
  constexpr int T = KS * KS, R = KS / 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = i % W, yy = (i / W) % H;
    const long bc = i / ((long)W * H);
    const int ci = bc % Cin; const long b = bc / Cin;
    float acc = 0.f;
    for (int co = 0; co < Cout; ++co) {
      const float* gp = dy + (b * Cout + co) * (long)H * W;
      const float* wp = w + ((long)co * Cin + ci) * T;
#pragma unroll
      for (int a = 0; a < KS; ++a) {
        const int r = yy - a + R;
        if (r < 0 || r >= H) continue;
#pragma unroll
        for (int c = 0; c < KS; ++c) {
          const int q = xx - c + R;
          if (q < 0 || q >= W) continue;
          acc += wp[a * KS + c] * gp[(long)r * W + q];
        }
      }
    }
    dx[i] = acc;
  }
}

// one workgroup per (co, ci) pair; threads stride over (b, pixel); all taps at once
template <int KS>
__global__ __launch_bounds__(256) void conv_direct_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dw, int B, int Cin, int Cout, int H, int W,
                                                         int accumulate) {
  constexpr int T = KS * KS, R = KS / 2;
  __shared__ float red[16];
  const int co = blockIdx.x / Cin, ci = blockIdx.x % Cin;
  float acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.f;
  const long HW = (long)H * W, P = (long)B * HW;
  for (long p = threadIdx.x; p < P; p += blockDim.x) {
    const long b = p / HW; const int pix = p % HW; const int yy = pix / W, xx = pix % W;
    const float g = dy[(b * Cout + co) * HW + pix];
    const float* xp = x + (b * Cin + ci) * HW;
#pragma unroll
    for (int a = 0; a < KS; ++a) {
      const int r = yy + a - R;
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        const int q = xx + c - R;
        if (r >= 0 && r < H && q >= 0 && q < W) acc[a * KS + c] += g * xp[(long)r * W + q];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float s = block_sum(acc[t], red);
    if (threadIdx.x == 0) {
      float* o = dw + ((long)co * Cin + ci) * T + t;
      *o = accumulate ? *o + s : s;
    }
  }
}

// dbias[n] = sum_{b,pix} dy[b,n,pix]: one wave per (b, n) plane writes part[b][n]; a second tiny kernel
// sums the B partials per channel (fixed order: deterministic)
__global__ __launch_bounds__(256) void conv_dbias_plane(const float* __restrict__ dy, float* __restrict__ part, long planes, int HW) {
  const long pl = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (pl >= planes) return;
  float s = 0.f;
  for (int i = lane; i < HW; i += 64) s += dy[pl * HW + i];
  s = wave_sum(s);
  if (lane == 0) part[pl] = s;
}
__global__ __launch_bounds__(256) void conv_dbias_final(const float* __restrict__ part, float* __restrict__ db, int B, int Cout, int accumulate) {
  // 256 threads = 32 channels x 8 batch groups
  __shared__ float red[8][33];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
  float s = 0.f;
  if (c < Cout) for (int b = g; b < B; b += 8) s += part[(long)b * Cout + c];
  red[g][threadIdx.x & 31] = s;
  __syncthreads();
  if (g == 0 && c < Cout) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31];
    db[c] = accumulate ? db[c] + t : t;
  }
}

// ------------------------------------------------------------------------------------------
// pixel-tile geometry shared by the MFMA kernels
// ------------------------------------------------------------------------------------------
// A tile is PT consecutive global pixels (global pixel = b*HW + y*W + x).  Either it is TR = PT/W
// whole rows of one image (HW % PT == 0, PT % W == 0) or TI = PT/HW whole images (PT % HW == 0).
struct TileGeom {
  int TI, TR;      // images per tile, rows per image in the tile
  int Wp;          // W + 2*halo
  int BS;          // LDS floats per (channel, image) block = (TR + 2*halo) * Wp
  int XS;          // LDS floats per channel
};
static inline bool tile_ok(int H, int W, int PT) {
  const int HW = H * W;
  return (HW % PT == 0 && PT % W == 0) || (PT % HW == 0);
}
static inline TileGeom make_geom(int H, int W, int PT, int halo) {
  TileGeom g;
  const int HW = H * W;
  if (HW >= PT) { g.TI = 1; g.TR = PT / W; } else { g.TI = PT / HW; g.TR = H; }
  g.Wp = W + 2 * halo;
  g.BS = (g.TR + 2 * halo) * g.Wp;
  g.XS = g.TI * g.BS;
  return g;
}

// LDS offset (within one channel) of tile pixel m's centre
__device__ __forceinline__ int pix_lds_off(int m, int W, const TileGeom& g, int halo) {
  const int per_img = g.TR * W;
  const int ti = m / per_img, r = (m % per_img) / W, c = m % W;
  return ti * g.BS + (r + halo) * g.Wp + (c + halo);
}

// For LDS position `pos` (0 <= pos < g.XS) of a tile starting at global pixel p0: the offset of the
// source element inside its (b, channel) plane and the image index, or -1 when the position is halo
// outside the image / beyond the batch.
__device__ __forceinline__ void pos_source(int pos, long p0, int H, int W, long P, const TileGeom& g, int halo,
                                           long& img, int& off) {
  const int HW = H * W;
  const int ti = pos / g.BS, rem = pos % g.BS;
  const int rr = rem / g.Wp - halo, cc = rem % g.Wp - halo;
  const long pimg = p0 / HW + ti;                   // image (batch index) of this block
  const int row0 = (int)((p0 % HW) / W);            // first tile row inside the image (0 when whole images)
  const int y = row0 + rr;
  img = pimg; off = -1;
  if (cc < 0 || cc >= W || y < 0 || y >= H) return;
  if (pimg * HW >= P) return;
  off = y * W + cc;
}

// ------------------------------------------------------------------------------------------
// forward / dgrad: D[n][pixel] = sum_{k, tap} Wt[n][k][tap] * X[k][pixel + tap]
//   FWD:   k = cin,  n = cout, w index ((n*K + k)*T + t)
//   DGRAD: k = cout, n = cin,  w index ((k*N + n)*T + (T-1-t))   (taps flipped)
// ------------------------------------------------------------------------------------------
// Register staging for conv_mfma_sk (generic form: addresses recomputed per chunk).
template <int T, bool DGRAD, int BN, int KC, int NP>
struct Stager {
  static constexpr int WS = KC * T + 1;
  static constexpr int NW = BN * KC * T / 256;
  static_assert(BN * KC * T % 256 == 0, "weight chunk must divide over 256 threads");
  float wreg[NW];
  float xreg[NP][KC];
  unsigned long long wok;       // bit e: weight element e is inside the tensor
  unsigned xok[NP];             // bit kc: channel k0+kc < K (and the position is not halo)

  __device__ __forceinline__ void fetch(const float* __restrict__ w, const float* __restrict__ x, int k0, int K, int N,
                                        int n0, int HW, const int (&spos)[NP], const int (&soff)[NP],
                                        const long (&simg)[NP], int XS) {
    wok = 0;
#pragma unroll
    for (int e = 0; e < NW; ++e) {
      const int i = threadIdx.x + 256 * e;
      int n, kc, t;
      if (!DGRAD) { n = i / (KC * T); const int r = i % (KC * T); kc = r / T; t = r % T; }
      else        { kc = i / (BN * T); const int r = i % (BN * T); n = r / T; t = r % T; }
      const bool ok = (n0 + n < N) && (k0 + kc < K);
      const int nc = min(n0 + n, N - 1), kk = min(k0 + kc, K - 1);
      wreg[e] = DGRAD ? w[(unsigned)((kk * N + nc) * T + t)] : w[(unsigned)((nc * K + kk) * T + t)];   // 32-bit offsets (host checks sizes)
      wok |= (ok ? 1ull : 0ull) << e;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      xok[p] = 0;
      if (spos[p] < XS) {
        const unsigned xb = soff[p] >= 0 ? (unsigned)simg[p] * (unsigned)(K * HW) + (unsigned)soff[p] : 0u;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          xreg[p][kc] = x[xb + (unsigned)(min(k0 + kc, K - 1) * HW)];
          xok[p] |= ((soff[p] >= 0 && k0 + kc < K) ? 1u : 0u) << kc;
        }
      }
    }
  }

  __device__ __forceinline__ void commit(float* __restrict__ Ws, float* __restrict__ Xs, const int (&spos)[NP], int XS) const {
#pragma unroll
    for (int e = 0; e < NW; ++e) {
      const int i = threadIdx.x + 256 * e;
      int dst;
      if (!DGRAD) { dst = (i / (KC * T)) * WS + i % (KC * T); }
      else        { const int kc = i / (BN * T), r = i % (BN * T); dst = (r / T) * WS + kc * T + r % T; }
      Ws[dst] = ((wok >> e) & 1ull) ? wreg[e] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p)
      if (spos[p] < XS) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) Xs[kc * XS + spos[p]] = ((xok[p] >> kc) & 1u) ? xreg[p][kc] : 0.f;
      }
  }
};

// compile-time tile geometry for the square maps of the UNet (PT = 128): {XS, Wp}; GEO 0 = run-time
template <int GEO> struct Geo { static constexpr int XS = 0, Wp = 0; };
template <> struct Geo<1> { static constexpr int XS = 204, Wp = 34; };   // 32x32: 4 rows      (6 x 34)
template <> struct Geo<2> { static constexpr int XS = 180, Wp = 18; };   // 16x16: 8 rows      (10 x 18)
template <> struct Geo<3> { static constexpr int XS = 200, Wp = 10; };   // 8x8:   2 images    (2 x 10 x 10)
template <> struct Geo<4> { static constexpr int XS = 288, Wp = 6; };    // 4x4:   8 images    (8 x 6 x 6)
template <> struct Geo<5> { static constexpr int XS = 128, Wp = 0; };    // 1x1 convolution: no halo, any map

// Main forward / dgrad kernel.
//  * per-thread staging PLAN (source offsets, LDS destinations, validity) is computed once; per K chunk
//    only a wave-uniform base moves, so a global load costs no vector ALU work;
//  * with a compile-time geometry every LDS fragment address is lane_base + immediate;
//  * two LDS buffers: chunk k+1 is written while chunk k is multiplied, ONE barrier per chunk, and the
//    global loads of chunk k+2 are in flight meanwhile.
template <int T, bool DGRAD, int BN, int KC, int GEO>
__global__ __launch_bounds__(256) void conv_mfma(const float* __restrict__ x, const float* __restrict__ w,
                                                 const float* __restrict__ bias, const float* __restrict__ res,
                                                 float* __restrict__ y, int B, int K, int N, int H, int W, int act,
                                                 TileGeom g) {
  constexpr int PT = 128, HALO = (T == 9) ? 1 : 0;
  constexpr int TM = (BN == 64) ? 2 : 1;            // 32-pixel MFMA tiles per wave (waves: 2x2 or 1x4)
  constexpr int WS = KC * T + 1;                    // odd row stride of the weight image
  constexpr int NP = (T == 9) ? 2 : 1;              // LDS positions per thread (XS <= 512 / <= 256)
  constexpr int NW = BN * KC * T / 256;
  static_assert(BN * KC * T % 256 == 0, "weight chunk must divide over 256 threads");
  const int XS = GEO ? Geo<GEO>::XS : g.XS;
  const int Wp = GEO ? Geo<GEO>::Wp : g.Wp;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int BUF = BN * WS + KC * XS;                 // floats per LDS buffer: [BN][WS] weights, [KC][XS] activations
  const int HW = H * W;
  const long P = (long)B * HW;
  const long p0 = (long)blockIdx.y * PT;
  const int n0 = blockIdx.x * BN;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nw = (BN == 64) ? (wv >> 1) * 32 : 0;                       // wave's first output channel in the tile
  const int mw = (BN == 64) ? (wv & 1) * 64 : wv * 32;                  // wave's first pixel in the tile
  const int half = lane >> 5, l31 = lane & 31;

  // ---- staging plan (chunk-invariant)
  unsigned wsrc[NW]; int wdst[NW]; unsigned wmask = 0;
  int wkc[NW];
#pragma unroll
  for (int e = 0; e < NW; ++e) {
    const int i = threadIdx.x + 256 * e;
    int n, kc, t;
    if (!DGRAD) { n = i / (KC * T); const int r = i % (KC * T); kc = r / T; t = r % T; }
    else        { kc = i / (BN * T); const int r = i % (BN * T); n = r / T; t = r % T; }
    const int nc = min(n0 + n, N - 1);
    wsrc[e] = DGRAD ? (unsigned)((kc * N + nc) * T + t) : (unsigned)(nc * K * T + kc * T + t);
    wdst[e] = n * WS + kc * T + t;
    wkc[e] = kc;
    wmask |= (n0 + n < N ? 1u : 0u) << e;
  }
  int spos[NP]; unsigned xsrc[NP]; bool xval[NP];
#pragma unroll
  for (int e = 0; e < NP; ++e) {
    spos[e] = threadIdx.x + 256 * e;
    int soff = -1; long simg = 0;
    if (spos[e] < XS) pos_source(spos[e], p0, H, W, P, g, HALO, simg, soff);
    xval[e] = soff >= 0;
    xsrc[e] = xval[e] ? (unsigned)simg * (unsigned)(K * HW) + (unsigned)soff : 0u;
  }
  const int wstep = DGRAD ? N * T : T;               // weight-offset advance per input channel
  int lbase[TM];                                     // lane part of the activation fragment address
#pragma unroll
  for (int mt = 0; mt < TM; ++mt) lbase[mt] = half * XS + pix_lds_off(mw + mt * 32 + l31, W, g, HALO);
  const int abase = (nw + l31) * WS + half * T;      // lane part of the weight fragment address

  float wreg[NW], xreg[NP][KC];
  auto fetch = [&](int k0) {
    const float* __restrict__ wk = w + (long)k0 * wstep;                // wave-uniform bases
    const float* __restrict__ xk = x + (long)k0 * HW;
    if (k0 + KC <= K) {
#pragma unroll
      for (int e = 0; e < NW; ++e) wreg[e] = wk[wsrc[e]];
#pragma unroll
      for (int p = 0; p < NP; ++p)
        if (spos[p] < XS) {
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) xreg[p][kc] = xk[xsrc[p] + (unsigned)(kc * HW)];
        }
    } else {                                         // ragged last chunk: clamp the channel, zero it at commit
#pragma unroll
      for (int e = 0; e < NW; ++e) wreg[e] = (k0 + wkc[e] < K) ? wk[wsrc[e]] : 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p)
        if (spos[p] < XS) {
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) xreg[p][kc] = (k0 + kc < K) ? xk[xsrc[p] + (unsigned)(kc * HW)] : 0.f;
        }
    }
  };
  auto commit = [&](float* __restrict__ buf) {
    float* __restrict__ Ws = buf;
    float* __restrict__ Xs = buf + BN * WS;
#pragma unroll
    for (int e = 0; e < NW; ++e) Ws[wdst[e]] = ((wmask >> e) & 1u) ? wreg[e] : 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      if (spos[p] < XS) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) Xs[kc * XS + spos[p]] = xval[p] ? xreg[p][kc] : 0.f;
      }
  };

  f32x16 acc[TM];
#pragma unroll
  for (int mt = 0; mt < TM; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  fetch(0);
  commit(smem);
  if (KC < K) fetch(KC);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < K; k0 += KC, cur ^= 1) {
    if (k0 + KC < K) {
      commit(smem + (cur ^ 1) * BUF);                 // chunk k0+KC (fetched one iteration ago) -> the idle buffer
      if (k0 + 2 * KC < K) fetch(k0 + 2 * KC);        // in flight during the MFMAs below
    }
    const float* __restrict__ Ws = smem + cur * BUF;
    const float* __restrict__ Xs = Ws + BN * WS;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int toff = (T == 9) ? ((t / 3 - 1) * Wp + (t % 3 - 1)) : 0;
      const int tw = DGRAD ? (T - 1 - t) : t;
#pragma unroll
      for (int k2 = 0; k2 < KC / 2; ++k2) {
        const float a = Ws[abase + 2 * k2 * T + tw];
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Xs[lbase[mt] + 2 * k2 * XS + toff], acc[mt], 0, 0, 0);
      }
    }
    __syncthreads();                                  // buffer `cur` is free again, buffer cur^1 is complete
  }
  // ---- epilogue: row n = (r&3) + 8*(r>>2) + 4*half, column = lane&31
#pragma unroll
  for (int mt = 0; mt < TM; ++mt) {
    const long p = p0 + mw + mt * 32 + l31;
    if (p >= P) continue;
    const long b = p / HW; const int pix = p % HW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + nw + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (n >= N) continue;
      const long idx = (b * N + n) * (long)HW + pix;
      float v = acc[mt][r];
      if (bias) v += bias[n];
      if (act == 1) v = gelu_erf(v);
      if (res) v += res[idx];
      y[idx] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward / dgrad for SMALL pixel counts (4x4 / 8x8 layers): 32 channels x 64 pixels per workgroup
// (4-8x more workgroups than the 64x128 tile), the four waves split every 32-channel K chunk four
// ways (8 channels each) and their accumulators meet in LDS at the end (in-workgroup split-K: no
// global partials, deterministic order).
// ------------------------------------------------------------------------------------------
template <int T, bool DGRAD>
__global__ __launch_bounds__(256) void conv_mfma_sk(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, const float* __restrict__ res,
                                                    float* __restrict__ y, int B, int K, int N, int H, int W, int act,
                                                    TileGeom g) {
  constexpr int PT = 64, BN = 32, KC = 32, KW = KC / 4, TM = 2, HALO = (T == 9) ? 1 : 0;
  constexpr int WS = KC * T + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                                  // [BN][WS]
  float* Xs = smem + BN * WS;                        // [KC][g.XS]
  const int HW = H * W;
  const long P = (long)B * HW;
  const long p0 = (long)blockIdx.y * PT;
  const int n0 = blockIdx.x * BN;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;

  int spos[1] = {(int)threadIdx.x}, soff[1] = {-1}; long simg[1] = {0};   // g.XS <= 256 on this path
  if (spos[0] < g.XS) pos_source(spos[0], p0, H, W, P, g, HALO, simg[0], soff[0]);
  int poff[TM];
#pragma unroll
  for (int mt = 0; mt < TM; ++mt) poff[mt] = pix_lds_off(mt * 32 + l31, W, g, HALO);

  f32x16 acc[TM];
#pragma unroll
  for (int mt = 0; mt < TM; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  Stager<T, DGRAD, BN, KC, 1> st;
  st.fetch(w, x, 0, K, N, n0, HW, spos, soff, simg, g.XS);
  for (int k0 = 0; k0 < K; k0 += KC) {
    __syncthreads();
    st.commit(Ws, Xs, spos, g.XS);
    __syncthreads();
    if (k0 + KC < K) st.fetch(w, x, k0 + KC, K, N, n0, HW, spos, soff, simg, g.XS);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int toff = (T == 9) ? ((t / 3 - 1) * g.Wp + (t % 3 - 1)) : 0;
      const int tw = DGRAD ? (T - 1 - t) : t;
#pragma unroll
      for (int k2 = 0; k2 < KW / 2; ++k2) {
        const int kc = wv * KW + 2 * k2 + half;
        const float a = Ws[l31 * WS + kc * T + tw];
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Xs[kc * g.XS + poff[mt] + toff], acc[mt], 0, 0, 0);
      }
    }
  }
  // ---- cross-wave reduction through LDS, then the epilogue (each thread owns 8 outputs)
  __syncthreads();
  float* red = smem;                                  // [4 waves][TM][16][64]
#pragma unroll
  for (int mt = 0; mt < TM; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wv * TM + mt) * 16 + r) * 64 + lane] = acc[mt][r];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < (TM * 16 * 64) / 256; ++e) {
    const int idx = threadIdx.x + 256 * e;
    const int mt = idx / 1024, r = (idx / 64) % 16, ln = idx % 64;
    float v = (red[idx] + red[idx + 2048]) + (red[idx + 4096] + red[idx + 6144]);
    const long p = p0 + mt * 32 + (ln & 31);
    const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
    if (p < P && n < N) {
      const long b = p / HW; const int pix = p % HW;
      const long o = (b * N + n) * (long)HW + pix;
      if (bias) v += bias[n];
      if (act == 1) v = gelu_erf(v);
      if (res) v += res[o];
      y[o] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// wgrad: part[slab][cout][cin][tap] = sum_{pixels of the slab} dY[cout][p] * X[cin][p + tap]
//   A = dY tile (rows = out channels, k = pixel), B = haloed X tile (columns = in channels), one
//   accumulator tile per tap.  A workgroup owns a (32*WNn couts) x (32*WCn cins) block and a range of
//   64-pixel chunks; its 4 waves are arranged WNn x WCn x WK: with fewer than 64 channels on a side the
//   spare waves split the pixels of every chunk (WK) and write their own slab, so no wave idles.
//   All 256 threads stage (lane = 32 consecutive tile positions -> 128-B reads); every load of a chunk
//   is issued before the first wait, and chunk k+1 is fetched while chunk k's 288 MFMAs run.
// ------------------------------------------------------------------------------------------
template <int T, int WNn, int WCn, int NPB, int NW>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_mfma(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ part, float* __restrict__ bias_part,
                                                       int B, int Cin, int Cout, int H, int W,
                                                       int chunks_per_split, int nchunks, TileGeom g) {
  constexpr int PT = 64, HALO = (T == 9) ? 1 : 0, S1 = PT + 1;
  constexpr int WK = NW / (WNn * WCn), BNo = 32 * WNn, BCi = 32 * WCn, NT_ = 64 * NW;
  constexpr int NG = BNo / NW;                       // dY rows per thread
  constexpr int NM = BCi / (2 * NW);                 // channels per thread per position slot
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int XSP = g.XS | 1;                          // odd channel stride: conflict-free fragment reads
  float* Gs = smem;                                  // [BNo][S1]   dY tile
  float* Xs = smem + BNo * S1;                       // [BCi][XSP]  X tile (haloed)
  int* Po = reinterpret_cast<int*>(Xs + BCi * XSP);  // [PT] pixel -> LDS offset
  const int HW = H * W;
  const long P = (long)B * HW;
  const int n0 = blockIdx.x * BNo, c0 = blockIdx.y * BCi, split = blockIdx.z;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wn = wv % WNn, wc = (wv / WNn) % WCn, wk = wv / (WNn * WCn);
  const int nw = wn * 32, cw = wc * 32;
  const int half = lane >> 5, l31 = lane & 31;
  const bool active = (n0 + nw < Cout) && (c0 + cw < Cin);             // wave-uniform

  if (threadIdx.x < PT) Po[threadIdx.x] = pix_lds_off(threadIdx.x, W, g, HALO);

  // ---- chunk-invariant staging plan
  const int pp = threadIdx.x & 63, ng = threadIdx.x >> 6;               // dY: pixel pp, rows ng + NW*e
  const int gdb = (HW >= PT) ? 0 : pp / HW;
  const int gpix = (HW >= PT) ? pp : pp % HW;
  const int px = threadIdx.x & 31, cg = threadIdx.x >> 5;               // X: position px + 32e, channels cg + 2*NW*m
  int p_ti[NPB], p_rr[NPB], p_cc[NPB];
#pragma unroll
  for (int e = 0; e < NPB; ++e) {
    const int pos = px + 32 * e;
    const int ti = pos / g.BS, rem = pos % g.BS;
    p_ti[e] = pos < g.XS ? ti : -1;
    p_rr[e] = rem / g.Wp - HALO;
    p_cc[e] = rem % g.Wp - HALO;
  }

  float gv[NG], xv[NPB][NM];
  unsigned gok, xok[NPB];
  // running origin of the next chunk to fetch: image, first pixel and first row inside it.  32-bit state advanced by
  // adds (the host bounds B*H*W*max(C) < 2^31) -- the per-chunk 64-bit divisions this replaces cost ~1 us each.
  int f_img, f_pix, f_row;
  auto fetch = [&]() {
    {  // dY
      const int b = f_img + gdb;
      const bool pin = b < B;
      const unsigned gb = pin ? (unsigned)(b * Cout) * (unsigned)HW + (unsigned)(f_pix + gpix) : 0u;
      gok = 0;
#pragma unroll
      for (int e = 0; e < NG; ++e) {
        const int n = n0 + ng + NW * e;
        gv[e] = dy[gb + (unsigned)(min(n, Cout - 1) * HW)];
        gok |= ((pin && n < Cout) ? 1u : 0u) << e;
      }
    }
#pragma unroll
    for (int e = 0; e < NPB; ++e) {
      xok[e] = 0;
      if (p_ti[e] >= 0) {
        const int img = f_img + p_ti[e];
        const int yy = f_row + p_rr[e];
        const bool ok = p_cc[e] >= 0 && p_cc[e] < W && yy >= 0 && yy < H && img < B;
        const unsigned xb = ok ? (unsigned)(img * Cin) * (unsigned)HW + (unsigned)(yy * W + p_cc[e]) : 0u;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          const int c = c0 + cg + 2 * NW * m;
          xv[e][m] = x[xb + (unsigned)(min(c, Cin - 1) * HW)];
          xok[e] |= ((ok && c < Cin) ? 1u : 0u) << m;
        }
      }
    }
    if (HW >= PT) {
      f_pix += PT; f_row += g.TR;
      if (f_pix >= HW) { f_pix = 0; f_row = 0; ++f_img; }
    } else {
      f_img += g.TI;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int e = 0; e < NG; ++e) Gs[(ng + NW * e) * S1 + pp] = ((gok >> e) & 1u) ? gv[e] : 0.f;
#pragma unroll
    for (int e = 0; e < NPB; ++e)
      if (p_ti[e] >= 0) {
#pragma unroll
        for (int m = 0; m < NM; ++m) Xs[(cg + 2 * NW * m) * XSP + px + 32 * e] = ((xok[e] >> m) & 1u) ? xv[e][m] : 0.f;
      }
  };

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const bool do_bias = bias_part != nullptr && blockIdx.y == 0;          // one cin-tile per (cout tile, split) sums dY
  float bsum = 0.f;

  const int cbeg = split * chunks_per_split;
  const int cend = min(nchunks, cbeg + chunks_per_split);
  {
    const unsigned pc0 = (unsigned)cbeg * PT;
    f_img = (int)(pc0 / (unsigned)HW);
    f_pix = (int)(pc0 % (unsigned)HW);
    f_row = f_pix / W;
  }
  if (cbeg < cend) fetch();
  for (int ch = cbeg; ch < cend; ++ch) {
    __syncthreads();
    commit();
    __syncthreads();
    if (ch + 1 < cend) fetch();                                   // in flight during the MFMAs
    if (do_bias) {                                                      // dbias: row sums of the dY tile (4 threads per row)
      const int bn = threadIdx.x >> 2, bq = threadIdx.x & 3;
      if (bn < BNo) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += Gs[bn * S1 + bq * 16 + j];
        bsum += t;
      }
    }
    if (active) {
#pragma unroll 4
      for (int s = wk * (PT / 2 / WK); s < (wk + 1) * (PT / 2 / WK); ++s) {
        const int pix = 2 * s + half;
        const float a = Gs[(nw + l31) * S1 + pix];
        const int xo = (cw + l31) * XSP + Po[pix];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int toff = (T == 9) ? ((t / 3 - 1) * g.Wp + (t % 3 - 1)) : 0;
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Xs[xo + toff], acc[t], 0, 0, 0);
        }
      }
    }
  }
  if (do_bias) {                                                        // slab tail: bias_part[split][Cout]
    float t = bsum;
    t += __shfl_xor(t, 1, kWave);
    t += __shfl_xor(t, 2, kWave);
    const int bn = threadIdx.x >> 2;
    if ((threadIdx.x & 3) == 0 && bn < BNo && n0 + bn < Cout) bias_part[(long)split * Cout + n0 + bn] = t;
  }
  float* out = part + (long)split * Cout * Cin * T;
  if (WK == 1) {
    if (!active) return;
    const int ci = c0 + cw + l31;
    if (ci < Cin) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + nw + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n >= Cout) continue;
#pragma unroll
        for (int t = 0; t < T; ++t) out[((long)t * Cout + n) * Cin + ci] = acc[t][r];      // slab layout [tap][cout][cin]: 128-B rows
      }
    }
  } else {
    // the WK waves that share an output block meet in LDS, one tap at a time (fixed order: deterministic)
    float* red = smem;                                  // [NW waves][16][64]
    constexpr int NB = WNn * WCn;                       // distinct output blocks in this workgroup (1 or 2)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wv * 16 + r) * 64 + lane] = acc[t][r];
      __syncthreads();
      for (int idx = threadIdx.x; idx < NB * 1024; idx += NT_) {
        const int blk = idx / 1024, r = (idx / 64) % 16, ln = idx % 64;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < WK; ++k) v += red[((k * NB + blk) * 16 + r) * 64 + ln];   // wave id = k*NB + blk
        const int bn = blk % WNn, bc = blk / WNn;
        const int n = n0 + bn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int ci = c0 + bc * 32 + (ln & 31);
        if (n < Cout && ci < Cin) out[((long)t * Cout + n) * Cin + ci] = v;
      }
    }
  }
}

// dw[i] (+)= sum_k part[k][i] and (optionally) db[j] (+)= sum_k bias_part[k][j] in ONE launch:
// 256 threads = 32 consecutive elements x 8 slab groups (fixed order); elements i >= n are the bias row.
// Slabs are [tap][cout][cin] (coalesced stores from the accumulator layout); dw is OIHW, so element i = (t, n*Cin+ci)
// lands at (n*Cin+ci)*T + t.
__global__ __launch_bounds__(256) void wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, long n, int splits,
                                                    const float* __restrict__ bias_part, float* __restrict__ db, int nb,
                                                    int accumulate, int T) {
  __shared__ float red[8][33];
  const long i = (long)blockIdx.x * 32 + (threadIdx.x & 31);
  const int g = threadIdx.x >> 5;
  const bool isw = i < n;
  const long j = isw ? i : i - n;
  const long stride = isw ? n : nb;
  const float* __restrict__ src = isw ? part : bias_part;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n + nb) {
    int k = g;
    for (; k + 24 < splits; k += 32) {
      s0 += src[(long)k * stride + j]; s1 += src[(long)(k + 8) * stride + j];
      s2 += src[(long)(k + 16) * stride + j]; s3 += src[(long)(k + 24) * stride + j];
    }
    for (; k < splits; k += 8) s0 += src[(long)k * stride + j];
  }
  red[g][threadIdx.x & 31] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && i < n + nb) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31];
    const long plane = n / T;
    float* o = isw ? dw + (j % plane) * T + j / plane : db + j;
    *o = accumulate ? *o + t : t;
  }
}

// the same reduction with 16-byte loads for n % 128 == 0 (every layer but the 3-channel ends): a workgroup owns 128
// consecutive elements x 8 slab groups, each half-wave streams a 512-byte run of one slab; blocks past n / 128 reduce
// the bias rows (32 per block) as above.  Fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce4(const float* __restrict__ part, float* __restrict__ dw, long n, int splits,
                                                     const float* __restrict__ bias_part, float* __restrict__ db, int nb,
                                                     int accumulate, int T) {
  __shared__ float4 red4[8][32];
  const int l = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long wblocks = n >> 7;
  if (blockIdx.x < wblocks) {
    const long j = ((long)blockIdx.x << 7) + 4 * l;
    const float4* __restrict__ src = reinterpret_cast<const float4*>(part + j);
    const long stride = n >> 2;
    float4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int k = g;
    for (; k + 24 < splits; k += 32) {
      const float4 a = src[(long)k * stride], b = src[(long)(k + 8) * stride], c = src[(long)(k + 16) * stride],
                   d = src[(long)(k + 24) * stride];
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
      s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
      s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
    }
    for (; k < splits; k += 8) {
      const float4 a = src[(long)k * stride];
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
    red4[g][l] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                             (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (threadIdx.x < 128) {                                  // one element per thread: (float4 slot e >> 2, component e & 3)
      const int e = threadIdx.x;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += reinterpret_cast<const float*>(&red4[q][e >> 2])[e & 3];
      const long je = ((long)blockIdx.x << 7) + e, plane = n / T;
      float* o = dw + (je % plane) * T + je / plane;
      *o = accumulate ? *o + t : t;
    }
  } else {
    __shared__ float red[8][33];
    const long j = (long)(blockIdx.x - wblocks) * 32 + l;
    float s0 = 0.f;
    if (j < nb)
      for (int k = g; k < splits; k += 8) s0 += bias_part[(long)k * nb + j];
    red[g][l] = s0;
    __syncthreads();
    if (g == 0 && j < nb) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += red[q][l];
      db[j] = accumulate ? db[j] + t : t;
    }
  }
}

int fold_launch(const afd_fold_desc* descs, int n, hipStream_t s);          // fold.hip

// The deterministic fold of weight-gradient slabs (and bias slabs) = one or two fold descriptors (fold.hip).  With `out`
// the descriptors are handed to the caller (who batches them with other layers' folds into one launch, later, on the
// same stream); without, they are folded now -- through the same kernel, so both ways give the same bits.
struct FoldSink { afd_fold_desc* out; int n; };
static int launch_wgrad_reduce(const float* part, float* dw, long n, int slabs, const float* bp, float* db, int nb, int accumulate,
                               int T, hipStream_t s, FoldSink* sink) {
  afd_fold_desc d[2];
  int nd = 0;
  d[nd++] = afd_fold_desc{part, dw, n, n, n / T, (long)T, slabs, accumulate};
  if (nb > 0) d[nd++] = afd_fold_desc{bp, db, (long)nb, (long)nb, (long)nb, 1L, slabs, accumulate};
  if (sink) {
    for (int i = 0; i < nd; ++i) sink->out[sink->n++] = d[i];
    return AFD_OK;
  }
  return fold_launch(d, nd, s);
}

// ------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------
// pw.hip: streaming kernel for the 1x1 projections
bool pw_launch(const float* in, const float* w, const float* bias, const float* res, float* out, int B, int K, int N, int P,
               int act, bool dgrad, hipStream_t s);
void pw_set_mode(int m);
// wino.hip: Winograd F(2x2,3x3) forward / dgrad
int wino_plan(int B, int K, int N, int H, int W);
bool wino_conv(const float* x, const float* w, const float* bias, const float* res, float* y, float* U, int B, int K, int N, int H,
               int W, int act, bool dgrad, bool weights_ready, hipStream_t s);
void wino_set_mode(int m);
void wino_weights_launch(const float* w, float* Uf, float* Ud, int Cin, int Cout, int kinds, hipStream_t s);
bool bf3_ok(int B, int K, int N, int H, int W);       // bf3.hip: the direct bf16x3 form of a 3x3 layer
void bf3_set_mode(int m);
void direct_form_set(int m);
bool direct_form_is_bf3();
void h2_sk_set_mode(int m);
void h2_lds_set(int m);
void wino_weights_batched_launch(const afd_wino_desc* descs, const int* wg_desc, int n_wg, hipStream_t s);
void wino_set_grid(int g);
int wgrad_wino_plan(int B, int Cin, int Cout, int H, int W, int* bn, int* bk, int* cps, int* nchunks);
// bf3_wgrad.hip: the 3x3 weight gradient on the bf16 matrix cores (three-piece splits)
int wgrad_bf3_plan(int B, int Cin, int Cout, int H, int W, int* tps, int* ntiles);
int wgrad_bf3(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int H, int W, hipStream_t s);
void wgrad_bf3_set_mode(int m);
void wgrad_arith_set(int m);
void wgrad_plan_target_set(long wgs);
bool wgrad_arith_is_bf3();
// h2_wgrad.hip: the same on the fp16 matrix cores (two-piece splits, online scaling; the plan is shared)
int wgrad_h2(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int H, int W, hipStream_t s);
int pw_wgrad_h2(const float* x, const float* dy, float* part, float* bias_part, int B, int Cin, int Cout, int L, hipStream_t s);
// ends.hip: the 1x1 output layer and its dgrad as streaming vector kernels
bool ends_fwd(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int Cin, int Cout, int H, int W,
              int ksize, int act, hipStream_t s);
bool ends_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H, int W, int ksize, hipStream_t s);
void ends_set_mode(int m);
int wgrad_cin3_plan(int B, int Cin, int Cout, int H, int W);
int wgrad_cin3(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int H, int W, hipStream_t s);
void pw_wgrad_bf3_set_mode(int m);
int pw_wgrad_bf3_plan(int B, int Cin, int Cout, int L, int* sps, int* nsteps, int* nr, int* nt);
int pw_wgrad_bf3(const float* x, const float* dy, float* part, float* bias_part, int B, int Cin, int Cout, int L, hipStream_t s);
int wgrad_wino(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int H, int W, hipStream_t s);
void wgrad_wino_set_mode(int m);

static inline bool use_mfma(int K, int N, int H, int W, int PT) { return N >= 8 && tile_ok(H, W, PT); }   // K < 8 (inc.conv1) rides the ragged-chunk path

static int g_conv_path = 0;       // 0 auto, 1 force the 64x128 tile, 2 force the split-K small tile (tests)

template <int T, bool DGRAD>
static int launch_mfma(const float* x, const float* w, const float* bias, const float* res, float* y,
                       int B, int K, int N, int H, int W, int act, hipStream_t s) {
  constexpr int KC = (T == 9) ? 8 : 32;
  const long P = (long)B * H * W;
  if (P * K >= (1L << 31) || (long)K * N * T >= (1L << 31)) return -1;   // staging uses 32-bit element offsets
  const unsigned ptiles = (unsigned)((P + 127) / 128);
  // tile choice by workgroup count (measured, tools/conv_paths.py): the 64x128 tile wins once it gives
  // every CU a workgroup; below that the 32x128 tile (twice the workgroups), and for the smallest
  // layers the 32x64 tile with in-workgroup split-K
  const long wgs64 = (long)ptiles * ((N + 63) / 64), wgs32 = (long)ptiles * ((N + 31) / 32);
  bool use32 = N <= 32 || (wgs64 < 256 && wgs32 >= 256);
  const bool want_sk = g_conv_path == 2 || (g_conv_path == 0 && wgs64 < 256 && wgs32 < 256);
  if (g_conv_path == 1) use32 = N <= 32;
  if (want_sk && K >= 32 && tile_ok(H, W, 64)) {         // (1x1 too: 128->128 @ 4x4 16.3 -> 12.4 us)
    const TileGeom g = make_geom(H, W, 64, T == 9 ? 1 : 0);
    if (g.XS <= 256) {
      const size_t lds_main = sizeof(float) * (32 * (32 * T + 1) + (size_t)32 * g.XS);
      const size_t lds_red = sizeof(float) * 4 * 2 * 16 * 64;
      const size_t lds = lds_main > lds_red ? lds_main : lds_red;
      hipLaunchKernelGGL((conv_mfma_sk<T, DGRAD>), dim3((N + 31) / 32, (unsigned)((P + 63) / 64)), dim3(256), lds, s,
                         x, w, bias, res, y, B, K, N, H, W, act, g);
      return 0;
    }
  }
  // (a 64x256 tile was measured too: 222 VGPRs -> 1 wave/SIMD, 5-9 % slower than 64x128 at 2 waves/SIMD)
  const TileGeom g = make_geom(H, W, 128, T == 9 ? 1 : 0);
  if (g.XS > 512) return -1;                                            // staging plan holds 2 positions per thread
  int geo = 0;
  if (T == 1) geo = 5;
  else if (H == W && (W == 32 || W == 16 || W == 8 || W == 4)) geo = W == 32 ? 1 : (W == 16 ? 2 : (W == 8 ? 3 : 4));
  const size_t lds = 2 * sizeof(float) * ((use32 ? 32 : 64) * (KC * T + 1) + (size_t)KC * g.XS);   // two buffers
  const dim3 grid(use32 ? (N + 31) / 32 : (N + 63) / 64, ptiles);
#define AFD_CONV_LAUNCH(BN_, GEO_) hipLaunchKernelGGL((conv_mfma<T, DGRAD, BN_, KC, GEO_>), grid, dim3(256), lds, s, x, w, bias, res, y, B, K, N, H, W, act, g)
#define AFD_CONV_GEO(BN_)                                             \
  switch (geo) {                                                       \
    case 1: if constexpr (T == 9) { AFD_CONV_LAUNCH(BN_, 1); } break;  \
    case 2: if constexpr (T == 9) { AFD_CONV_LAUNCH(BN_, 2); } break;  \
    case 3: if constexpr (T == 9) { AFD_CONV_LAUNCH(BN_, 3); } break;  \
    case 4: if constexpr (T == 9) { AFD_CONV_LAUNCH(BN_, 4); } break;  \
    case 5: if constexpr (T == 1) { AFD_CONV_LAUNCH(BN_, 5); } break;  \
    default: AFD_CONV_LAUNCH(BN_, 0);                                  \
  }
  if (!use32) { AFD_CONV_GEO(64) } else { AFD_CONV_GEO(32) }
#undef AFD_CONV_GEO
#undef AFD_CONV_LAUNCH
  return 0;
}

static int g_wgrad_nw = 0;        // tuning hook: waves per wgrad workgroup (0 = default 8)
struct WgradPlan { int wnn, wcn, wk, splits, slabs, nchunks, cps, npb; };
static inline WgradPlan wgrad_plan(int B, int Cin, int Cout, int H, int W, int ksize) {
  WgradPlan p;
  p.wnn = Cout > 32 ? 2 : 1;
  p.wcn = Cin > 32 ? 2 : 1;
  p.wk = 4 / (p.wnn * p.wcn);
  p.nchunks = (int)(((long)B * H * W + 63) / 64);
  const long tiles = (long)((Cout + 32 * p.wnn - 1) / (32 * p.wnn)) * ((Cin + 32 * p.wcn - 1) / (32 * p.wcn));
  static const long target = [] { const char* e = getenv("AFD_WG_TARGET"); return e ? atol(e) : 256L; }();   // tuning hook
  long s = target / tiles;                  // one workgroup per CU: the kernel is register-heavy (1 wave/SIMD)
  if (s < 1) s = 1;
  if (s > p.nchunks) s = p.nchunks;
  p.cps = (int)((p.nchunks + s - 1) / s);
  p.splits = (int)((p.nchunks + p.cps - 1) / p.cps);   // no empty splits
  p.slabs = p.splits;                                  // pixel-split waves meet in LDS: one slab per split
  const TileGeom g = make_geom(H, W, 64, ksize == 3 ? 1 : 0);
  const int need = (g.XS + 31) / 32;
  p.npb = ksize == 1 ? 2 : (need <= 4 ? 4 : (need <= 5 ? 5 : 8));
  return p;
}

template <int T, int WNn, int WCn, int NPB>
static void launch_wgrad(const float* x, const float* dy, float* part, float* bias_part, int B, int Cin, int Cout, int H, int W,
                         const WgradPlan& p, const TileGeom& g, hipStream_t s) {
  // 8 waves (two per SIMD: one wave's staging and epilogue overlap the other's MFMAs, same slab count) pay off once a
  // workgroup has several chunks to stream; measured per layer in tools/wgrad_nw.py
  const int nw = g_wgrad_nw ? g_wgrad_nw : ((p.cps >= 8 || (p.cps >= 4 && H * W >= 256)) ? 8 : 4);
  size_t lds = sizeof(float) * ((size_t)32 * WNn * 65 + (size_t)32 * WCn * (g.XS | 1)) + sizeof(int) * 64;
  const size_t red = sizeof(float) * nw * 16 * 64;                                                // cross-wave reduce buffer
  if (WNn * WCn < nw && lds < red) lds = red;
  const dim3 grid((Cout + 32 * WNn - 1) / (32 * WNn), (Cin + 32 * WCn - 1) / (32 * WCn), p.splits);
  if (nw == 8)
    hipLaunchKernelGGL((conv_wgrad_mfma<T, WNn, WCn, NPB, 8>), grid, dim3(512), lds, s, x, dy, part, bias_part, B, Cin, Cout, H, W, p.cps, p.nchunks, g);
  else
    hipLaunchKernelGGL((conv_wgrad_mfma<T, WNn, WCn, NPB, 4>), grid, dim3(256), lds, s, x, dy, part, bias_part, B, Cin, Cout, H, W, p.cps, p.nchunks, g);
}

template <int T, int NPB>
static void launch_wgrad_roles(const float* x, const float* dy, float* part, float* bp, int B, int Cin, int Cout, int H, int W,
                               const WgradPlan& p, const TileGeom& g, hipStream_t s) {
  if (p.wnn == 2 && p.wcn == 2) launch_wgrad<T, 2, 2, NPB>(x, dy, part, bp, B, Cin, Cout, H, W, p, g, s);
  else if (p.wnn == 2) launch_wgrad<T, 2, 1, NPB>(x, dy, part, bp, B, Cin, Cout, H, W, p, g, s);
  else if (p.wcn == 2) launch_wgrad<T, 1, 2, NPB>(x, dy, part, bp, B, Cin, Cout, H, W, p, g, s);
  else launch_wgrad<T, 1, 1, NPB>(x, dy, part, bp, B, Cin, Cout, H, W, p, g, s);
}

}  // namespace afd
using namespace afd;

extern "C" {

int afd_debug_conv_path(int mode) {
  if (mode >= 92 && mode <= 93) { ends_set_mode(mode - 92); return AFD_OK; }   // output-layer streaming kernels: 92 = by rule (default), 93 = off
  if (mode >= 88 && mode <= 89) { pw_wgrad_bf3_set_mode(mode - 88); return AFD_OK; }   // bf16x3 1x1 wgrad: 88 = by rule (default), 89 = off
  if (mode >= 84 && mode <= 86) { wgrad_bf3_set_mode(mode - 84); return AFD_OK; }   // bf16x3 3x3 wgrad: 84 = by rule (default), 85 = off, 86 = wherever covered
  if (mode >= 48 && mode <= 49) { wgrad_plan_target_set(mode == 48 ? 256 : 160); return AFD_OK; }   // 3x3 matrix-core weight gradient, workgroups per launch: 48 = one per CU (fastest alone; default), 49 = 160 (fastest beside the dependent chain of a train step: TrainStep asks for it)
  if (mode >= 59 && mode <= 63) { h2_lds_set(mode == 59 ? 4 : (mode == 60 ? 1 : (mode == 61 ? 0 : mode - 60))); return AFD_OK; }   // f16x2 tile kernel: 60 = both operands from LDS, weights by LDS-DMA (default), 61 = the register-fed kernel (before round 3's last third), 62 / 63 / 59 = LDS-fed wherever covered, the (128 px, 64 ch) / (256 px, 32 ch) / (128 px, 32 ch) workgroup first (tests)
  if (mode == 73) { h2_sk_set_mode(2); return AFD_OK; }                          // ... 73 = wherever the shape is covered (tests)
  if (mode >= 74 && mode <= 75) { h2_sk_set_mode(mode - 74); return AFD_OK; }    // f16x2 split-K kernel for the 4x4 / thin 8x8 maps: 74 = by rule (default), 75 = off (round 1's fp32 Winograd split-K kernel)
  if (mode >= 78 && mode <= 79) { wgrad_arith_set(mode - 78); return AFD_OK; }   // arithmetic of the matrix-core 3x3 weight gradient: 78 = f16x2 (default), 79 = bf16x3 (round 2)
  if (mode >= 76 && mode <= 77) { direct_form_set(mode - 76); return AFD_OK; }   // arithmetic of the direct 3x3 forward / dgrad: 76 = f16x2 (default), 77 = bf16x3 (round 2)
  if (mode >= 80 && mode <= 82) { bf3_set_mode(mode - 80); return AFD_OK; }   // direct bf16x3 3x3 kernel: 80 = by rule (default), 81 = off, 82 = wherever covered
  if (mode >= 8 && mode <= 10) { pw_set_mode(mode - 8); return AFD_OK; }     // 1x1 streaming kernel: 8 = by rule (default), 9 = off, 10 = forced
  if (mode >= 100000 && mode < 200000) { wino_set_grid(mode - 100000); return AFD_OK; }   // Winograd persistent grid size (0 = default)
  if (mode >= 96 && mode <= 98) { wgrad_wino_set_mode(mode - 96); return AFD_OK; }   // Winograd wgrad: 96 = by rule (default), 97 = off, 98 = whenever covered
  if (mode >= 64 && mode <= 70) { wino_set_mode(mode - 64); return AFD_OK; }   // Winograd 3x3: 64 = by rule (default), 65 = off, 66..69 = forced, workgroups of 64x64 / 32x64 / 64x32 / 32x32 (channels x tiles)
  if (mode >= 32 && mode <= 34) { g_wgrad_nw = mode == 32 ? 4 : (mode == 33 ? 8 : 0); return AFD_OK; }   // wgrad waves: 4 / 8 / auto
  AFD_REQUIRE(mode >= 0 && mode <= 2, "afd_debug_conv_path: mode must be 0 (auto), 1 (big tile), 2 (split-K tile), 8..10 (1x1 streaming auto/off/forced), 32..34 (wgrad waves 4/8/auto), 64..69 (Winograd fwd/dgrad auto/off/forced shapes) or 96..98 (Winograd wgrad auto/off/forced)");
  g_conv_path = mode;
  return AFD_OK;
}

int afd_conv_fwd(const float* x, const float* w, const float* bias, const float* res, float* y,
                 int B, int Cin, int Cout, int H, int W, int ksize, int act, afd_stream_t st) {
  AFD_REQUIRE(x && w && y && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "afd_conv_fwd: bad argument");
  AFD_REQUIRE(ksize == 1 || ksize == 3, "afd_conv_fwd: ksize %d not in {1,3}", ksize);
  AFD_REQUIRE(act == 0 || act == 1, "afd_conv_fwd: act must be 0 or 1");
  hipStream_t s = as_stream(st);
  if (ends_fwd(x, w, bias, res, y, B, Cin, Cout, H, W, ksize, act, s)) return check_launch("afd_conv_fwd");
  if (ksize == 1 && pw_launch(x, w, bias, res, y, B, Cin, Cout, H * W, act, false, s)) return check_launch("afd_conv_fwd");
  int rc = -1;
  if (use_mfma(Cin, Cout, H, W, 128))
    rc = ksize == 3 ? launch_mfma<9, false>(x, w, bias, res, y, B, Cin, Cout, H, W, act, s)
                    : launch_mfma<1, false>(x, w, bias, res, y, B, Cin, Cout, H, W, act, s);
  if (rc != 0) {
    const long total = (long)B * Cout * H * W;
    if (ksize == 3) hipLaunchKernelGGL(conv_direct_fwd<3>, dim3(gs_grid(total)), dim3(256), 0, s, x, w, bias, res, y, Cin, Cout, H, W, act, total);
    else hipLaunchKernelGGL(conv_direct_fwd<1>, dim3(gs_grid(total)), dim3(256), 0, s, x, w, bias, res, y, Cin, Cout, H, W, act, total);
  }
  return check_launch("afd_conv_fwd");
}

int afd_conv_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H, int W, int ksize, afd_stream_t st) {
  AFD_REQUIRE(dy && w && dx && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "afd_conv_dgrad: bad argument");
  AFD_REQUIRE(ksize == 1 || ksize == 3, "afd_conv_dgrad: ksize %d not in {1,3}", ksize);
  hipStream_t s = as_stream(st);
  if (ends_dgrad(dy, w, dx, B, Cin, Cout, H, W, ksize, s)) return check_launch("afd_conv_dgrad");
  if (ksize == 1 && pw_launch(dy, w, nullptr, nullptr, dx, B, Cout, Cin, H * W, 0, true, s)) return check_launch("afd_conv_dgrad");
  int rc = -1;
  if (use_mfma(Cout, Cin, H, W, 128))     // reduction over Cout, output channels = Cin
    rc = ksize == 3 ? launch_mfma<9, true>(dy, w, nullptr, nullptr, dx, B, Cout, Cin, H, W, 0, s)
                    : launch_mfma<1, true>(dy, w, nullptr, nullptr, dx, B, Cout, Cin, H, W, 0, s);
  if (rc != 0) {
    const long total = (long)B * Cin * H * W;
    if (ksize == 3) hipLaunchKernelGGL(conv_direct_dgrad<3>, dim3(gs_grid(total)), dim3(256), 0, s, dy, w, dx, Cin, Cout, H, W, total);
    else hipLaunchKernelGGL(conv_direct_dgrad<1>, dim3(gs_grid(total)), dim3(256), 0, s, dy, w, dx, Cin, Cout, H, W, total);
  }
  return check_launch("afd_conv_dgrad");
}

size_t afd_conv3x3_wino_workspace_bytes(int B, int Cin, int Cout, int H, int W, int dgrad) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
  // (the bf16x3 weight image needs 54 bytes per (cin, cout) pair, the Winograd one 64: one size serves both)
  return (bf3_ok(B, K, N, H, W) || wino_plan(B, K, N, H, W)) ? sizeof(float) * 16 * (size_t)Cin * Cout : 0;
}

int afd_conv3x3_weight_kinds(int B, int Cin, int Cout, int H, int W) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  const int k = (bf3_ok(B, Cin, Cout, H, W) ? 1 : 0) | (bf3_ok(B, Cout, Cin, H, W) ? 2 : 0);
  return k && direct_form_is_bf3() ? k | 4 : k;
}

int afd_conv3x3_wino_weights(const float* w, float* u_fwd, float* u_dgrad, int Cin, int Cout, int kinds, afd_stream_t st) {
  AFD_REQUIRE(w && (u_fwd || u_dgrad) && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, "afd_conv3x3_wino_weights: bad argument");
  AFD_REQUIRE(kinds >= 0 && kinds <= 7 && (!(kinds & 4) || (kinds & 3)) && (!(kinds & 3) || (Cin % 16 == 0 && Cout % 16 == 0)), "afd_conv3x3_wino_weights: bad kinds %d", kinds);
  wino_weights_launch(w, u_fwd, u_dgrad, Cin, Cout, kinds, as_stream(st));
  return check_launch("afd_conv3x3_wino_weights");
}

int afd_conv3x3_wino_weights_batched(const afd_wino_desc* descs, const int* wg_desc, int n_wg, afd_stream_t st) {
  AFD_REQUIRE(descs && wg_desc && n_wg > 0, "afd_conv3x3_wino_weights_batched: bad argument");
  wino_weights_batched_launch(descs, wg_desc, n_wg, as_stream(st));
  return check_launch("afd_conv3x3_wino_weights_batched");
}

// the weight image in the workspace was built for `kinds` (afd_conv3x3_weight_kinds at build time): it must be the form this
// call is about to read -- the rule depends on the batch size and on the debug switches
static int check_kinds(const char* who, int kinds, int B, int Cin, int Cout, int H, int W, int dgrad) {
  const int now = afd_conv3x3_weight_kinds(B, Cin, Cout, H, W);
  const int bit = dgrad ? 2 : 1;
  AFD_REQUIRE(kinds >= 0 && (kinds & bit) == (now & bit) && (!(now & bit) || (kinds & 4) == (now & 4)),
              "%s: the weight image was built as kinds %d, but a call with (B %d, %d -> %d, %dx%d) under the current settings reads kinds %d "
              "(rebuild it: afd_conv3x3_weight_kinds depends on the batch size and on afd_debug_conv_path)", who, kinds, B, Cin, Cout, H, W, now);
  return AFD_OK;
}

int afd_conv3x3_wino_fwd(const float* x, const float* w, const float* bias, const float* res, float* y,
                         int B, int Cin, int Cout, int H, int W, int act, void* workspace, int weights_ready, int kinds, afd_stream_t st) {
  AFD_REQUIRE(x && w && y && workspace && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "afd_conv3x3_wino_fwd: bad argument");
  AFD_REQUIRE(act == 0 || act == 1, "afd_conv3x3_wino_fwd: act must be 0 or 1");
  if (weights_ready && check_kinds("afd_conv3x3_wino_fwd", kinds, B, Cin, Cout, H, W, 0) != AFD_OK) return AFD_EINVAL;
  AFD_REQUIRE(wino_conv(x, w, bias, res, y, static_cast<float*>(workspace), B, Cin, Cout, H, W, act, false, weights_ready != 0, as_stream(st)),
              "afd_conv3x3_wino_fwd: shape (%d,%d->%d,%dx%d) is not covered (afd_conv3x3_wino_workspace_bytes returns 0 for it)", B, Cin, Cout, H, W);
  return check_launch("afd_conv3x3_wino_fwd");
}

int afd_conv3x3_wino_dgrad(const float* dy, const float* w, float* dx, const float* add_to_dx, int B, int Cin, int Cout, int H, int W,
                           void* workspace, int weights_ready, int kinds, afd_stream_t st) {
  AFD_REQUIRE(dy && w && dx && workspace && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "afd_conv3x3_wino_dgrad: bad argument");
  if (weights_ready && check_kinds("afd_conv3x3_wino_dgrad", kinds, B, Cin, Cout, H, W, 1) != AFD_OK) return AFD_EINVAL;
  AFD_REQUIRE(wino_conv(dy, w, nullptr, add_to_dx, dx, static_cast<float*>(workspace), B, Cout, Cin, H, W, 0, true, weights_ready != 0, as_stream(st)),
              "afd_conv3x3_wino_dgrad: shape (%d,%d->%d,%dx%d) is not covered", B, Cin, Cout, H, W);
  return check_launch("afd_conv3x3_wino_dgrad");
}

size_t afd_conv_wgrad_workspace_bytes(int B, int Cin, int Cout, int H, int W, int ksize) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  size_t pw = 0;                                                          // the 1x1 bf16x3 form: [slabs][Cout*Cin] + [slabs][Cout]
  if (ksize == 1) {
    int a, b, c, d;
    pw = sizeof(float) * (size_t)pw_wgrad_bf3_plan(B, Cin, Cout, H * W, &a, &b, &c, &d) * ((size_t)Cout * Cin + Cout);
  }
  if (!tile_ok(H, W, 64)) return std::max(pw, sizeof(float) * (size_t)B * Cout);
  // [slabs][Cout*Cin*T] weight slabs, then [slabs][Cout] bias slabs; the direct path wants (B, Cout) plane sums
  size_t slabs = wgrad_plan(B, Cin, Cout, H, W, ksize).slabs;
  if (ksize == 3) {                                                      // the Winograd / bf16x3 forms may split finer
    int bn, bk, cps, nch;
    const size_t ws = (size_t)wgrad_wino_plan(B, Cin, Cout, H, W, &bn, &bk, &cps, &nch);
    if (ws > slabs) slabs = ws;
    const size_t wb = (size_t)wgrad_bf3_plan(B, Cin, Cout, H, W, &cps, &nch);
    if (wb > slabs) slabs = wb;
    const size_t wc = (size_t)wgrad_cin3_plan(B, Cin, Cout, H, W);
    if (wc > slabs) slabs = wc;
  }
  const size_t need = slabs * ((size_t)Cout * Cin * ksize * ksize + Cout);
  const size_t direct = (size_t)B * Cout;
  return std::max(pw, sizeof(float) * (need > direct ? need : direct));
}

int afd_conv_wgrad_form(int B, int Cin, int Cout, int H, int W, int ksize) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  int a, b, c, d;
  if (ksize == 1) return pw_wgrad_bf3_plan(B, Cin, Cout, H * W, &a, &b, &c, &d) ? (wgrad_arith_is_bf3() ? 2 : 4) : 0;
  if (ksize != 3) return 0;
  if (wgrad_bf3_plan(B, Cin, Cout, H, W, &a, &b)) return wgrad_arith_is_bf3() ? 2 : 4;
  if (wgrad_cin3_plan(B, Cin, Cout, H, W)) return 3;
  return wgrad_wino_plan(B, Cin, Cout, H, W, &a, &b, &c, &d) ? 1 : 0;
}

static int conv_wgrad_impl(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout, int H, int W,
                           int ksize, int accumulate, void* workspace, FoldSink* sink, hipStream_t s, const char* who) {
  AFD_REQUIRE(x && dy && dw && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad argument", who);
  AFD_REQUIRE(ksize == 1 || ksize == 3, "%s: ksize %d not in {1,3}", who, ksize);
  const int T = ksize * ksize;
  const TileGeom g = make_geom(H, W, 64, ksize == 3 ? 1 : 0);
  int rc = AFD_OK;
  if (ksize == 3 && !dbias && workspace) {                               // bf16x3 form on the matrix pipe, else Winograd F(3x3, 2x2): 16 multiplies per tile instead of 36
    float* part = static_cast<float*>(workspace);
    int slabs = wgrad_arith_is_bf3() ? wgrad_bf3(x, dy, part, B, Cin, Cout, H, W, s) : wgrad_h2(x, dy, part, B, Cin, Cout, H, W, s);
    if (!slabs) slabs = wgrad_cin3(x, dy, part, B, Cin, Cout, H, W, s);
    if (!slabs) slabs = wgrad_wino(x, dy, part, B, Cin, Cout, H, W, s);
    if (slabs) {
      const long n = (long)Cout * Cin * 9;
      rc = launch_wgrad_reduce(part, dw, n, slabs, nullptr, nullptr, 0, accumulate, 9, s, sink);
      return rc != AFD_OK ? rc : check_launch(who);
    }
  }
  if (ksize == 1 && workspace) {                                         // bf16x3 form straight from global memory (HBM-bound)
    int a, b, c, d;
    const int slabs = pw_wgrad_bf3_plan(B, Cin, Cout, H * W, &a, &b, &c, &d);
    if (slabs) {
      float* part = static_cast<float*>(workspace);
      const long n = (long)Cout * Cin;
      float* bp = dbias ? part + (size_t)slabs * n : nullptr;
      if (wgrad_arith_is_bf3()) pw_wgrad_bf3(x, dy, part, bp, B, Cin, Cout, H * W, s);
      else pw_wgrad_h2(x, dy, part, bp, B, Cin, Cout, H * W, s);
      rc = launch_wgrad_reduce(part, dw, n, slabs, bp, dbias, dbias ? Cout : 0, accumulate, 1, s, sink);
      return rc != AFD_OK ? rc : check_launch(who);
    }
  }
  if (tile_ok(H, W, 64) && g.XS <= 256 && (long)B * H * W * (Cin > Cout ? Cin : Cout) < (1L << 31)) {
    AFD_REQUIRE(workspace, "%s: workspace is NULL", who);
    const WgradPlan p = wgrad_plan(B, Cin, Cout, H, W, ksize);
    float* part = static_cast<float*>(workspace);
    const long n = (long)Cout * Cin * T;
    float* bp = dbias ? part + (size_t)p.slabs * n : nullptr;           // bias slabs behind the weight slabs
    if (ksize == 1) launch_wgrad_roles<1, 2>(x, dy, part, bp, B, Cin, Cout, H, W, p, g, s);
    else if (p.npb == 4) launch_wgrad_roles<9, 4>(x, dy, part, bp, B, Cin, Cout, H, W, p, g, s);
    else if (p.npb == 5) launch_wgrad_roles<9, 5>(x, dy, part, bp, B, Cin, Cout, H, W, p, g, s);
    else launch_wgrad_roles<9, 8>(x, dy, part, bp, B, Cin, Cout, H, W, p, g, s);
    rc = launch_wgrad_reduce(part, dw, n, p.slabs, bp, dbias, dbias ? Cout : 0, accumulate, T, s, sink);
    return rc != AFD_OK ? rc : check_launch(who);
  } else {
    const dim3 grid((unsigned)(Cout * Cin));
    if (ksize == 3) hipLaunchKernelGGL(conv_direct_wgrad<3>, grid, dim3(256), 0, s, x, dy, dw, B, Cin, Cout, H, W, accumulate);
    else hipLaunchKernelGGL(conv_direct_wgrad<1>, grid, dim3(256), 0, s, x, dy, dw, B, Cin, Cout, H, W, accumulate);
  }
  if (dbias) {                                                            // direct path only: plane sums + column sum
    AFD_REQUIRE(workspace, "%s: workspace is NULL", who);
    float* bp = static_cast<float*>(workspace);
    const long planes = (long)B * Cout;
    hipLaunchKernelGGL(conv_dbias_plane, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, s, dy, bp, planes, H * W);
    hipLaunchKernelGGL(conv_dbias_final, dim3((Cout + 31) / 32), dim3(256), 0, s, bp, dbias, B, Cout, accumulate);
  }
  return check_launch(who);
}

int afd_conv_wgrad(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout, int H, int W,
                   int ksize, int accumulate, void* workspace, afd_stream_t st) {
  return conv_wgrad_impl(x, dy, dw, dbias, B, Cin, Cout, H, W, ksize, accumulate, workspace, nullptr, as_stream(st), "afd_conv_wgrad");
}

int afd_conv_wgrad_partials(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout, int H, int W,
                            int ksize, int accumulate, void* workspace, afd_fold_desc* folds_out, int* n_folds, afd_stream_t st) {
  AFD_REQUIRE(folds_out && n_folds, "afd_conv_wgrad_partials: folds_out / n_folds are NULL");
  FoldSink sink{folds_out, 0};
  const int rc = conv_wgrad_impl(x, dy, dw, dbias, B, Cin, Cout, H, W, ksize, accumulate, workspace, &sink, as_stream(st), "afd_conv_wgrad_partials");
  *n_folds = sink.n;
  return rc;
}

}  // extern "C"

// h2_wgrad.hip -- F5: weight gradient of the 3x3 (pad 1) convolution on the fp16 matrix cores with TWO-piece splits and
// online power-of-two scaling (round 3; h2_common.h has the arithmetic and its error bound).  Same reduction-over-pixels
// GEMM and the same tiling as round 2's bf16x3 kernel (bf3_wgrad.hip, kept for A/B: afd_debug_conv_path 78 / 79):
//
//   dW[n][k][ty][tx] = sum_q dY[n][q - (tx-1)] * X[k][q + (ty-1) W],       dY taken as 0 outside its row
//
//   B operand (X):  staged once per 128-pixel tile, scaled by the WORKGROUP's running scale s_x, split, kept in LDS as 16-byte
//                   records [piece 2][16-channel block][group of 8 pixels, halo rows zero][channel];
//   A operand (dY): each wave reads ITS 16 output channels straight from global memory, scales them by the WAVE's running scale
//                   s_d, splits the 10 values once and packs the three column-shifted fragments in registers.
//   NINE v_mfma_f32_16x16x32_f16 per (16-channel block, row shift) unit instead of eighteen.  Both scales follow the data
//   (lowered when a tile / a 32-pixel step would overflow fp16, the accumulators multiplied by the ratio); before the wave
//   groups are summed and the slab is written every wave multiplies its sums by 1 / (s_x s_d).  All factors are powers of two.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include "common.h"
#include "h2_common.h"

namespace afd {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

template <int S, int BN_, int BK_> struct HwGeo {
  static constexpr int TP = 128, G8 = S / 8;                          // pixels per tile, 8-pixel groups per row
  static constexpr int IPT = S * S >= TP ? 1 : TP / (S * S);          // images per tile
  static constexpr int R = S * S >= TP ? TP / S : S;                  // rows per image in the tile
  static constexpr int TPI = S * S >= TP ? S * S / TP : 1;            // tiles per image
  static constexpr int IG = (R + 2) * G8, NG = IPT * IG;              // groups per image incl. the two halo rows / per tile
  static constexpr int BK = BK_, BN = BN_;
  static constexpr int NR = (BN / 16) * (BK / 32), NSG = 8 / NR;      // wave roles, wave groups sharing a tile's steps
  static constexpr int TASKS = (BK / 16) * NG * 16, NE = (TASKS + 511) / 512;
  static constexpr int PIECE = (BK / 16) * NG * 16;                   // records per piece
};

__device__ __forceinline__ uint32_t h_pack(_Float16 lo, _Float16 hi) {
  return (uint32_t)__builtin_bit_cast(uint16_t, lo) | ((uint32_t)__builtin_bit_cast(uint16_t, hi) << 16);
}

// wave groups 1 .. NSG-1 hand their sums to group 0 through LDS, in a fixed order; group 0 writes the slab [split][tap][cout][cin]
// (accumulator: row n = 4 (l >> 4) + reg, column k = l & 15)
template <int NR, int NSG>
__device__ __forceinline__ void hw_finish(f32x4 (&acc)[2][9], float unscale_x, float unscale_d, uint8_t* smem_raw, float* __restrict__ part,
                                          int split, int role, int sg, int lane, int n0w, int k0w, int N, int K) {
  const int l15 = lane & 15, kgl = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[ks][t] = (acc[ks][t] * unscale_x) * unscale_d;      // the true sums: every wave had its own s_d
  if (NSG > 1) {
    float* red = reinterpret_cast<float*>(smem_raw);                  // [role][72][64]
#pragma unroll 1
    for (int g = 1; g < NSG; ++g) {
      __syncthreads();                                                 // (first round: the last tile's fragment reads are done)
      if (sg == g) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) red[(role * 72 + (ks * 9 + t) * 4 + rg) * 64 + lane] = acc[ks][t][rg];
      }
      __syncthreads();
      if (sg == 0) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) acc[ks][t][rg] += red[(role * 72 + (ks * 9 + t) * 4 + rg) * 64 + lane];
      }
    }
    if (sg != 0) return;
  }
  float* ps = part + (long)split * 9 * N * K;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) ps[((long)t * N + n0w + 4 * kgl + rg) * K + k0w + 16 * ks + l15] = acc[ks][t][rg];
}

// the 27 + 27 MFMAs of one 32-pixel step: units (16-channel block ks, row shift ty), B fragments one unit ahead; gb = record of
// (piece 0, this wave's first channel block, the lane's group at row shift 0, channel l & 15), kstride = records per channel
// block, toff[ty] = record offset of row shift ty
__device__ __forceinline__ void hw_step(f32x4 (&acc)[2][9], const h8 (&af)[3][2], const h8* __restrict__ Xs, int PIECE, int gb, int kstride,
                                        int toff0, int toff1, int toff2) {
  h8 bc[2], bn[2];
  { const int o = gb + toff0; bc[0] = Xs[o]; bc[1] = Xs[PIECE + o]; }
#pragma unroll
  for (int u = 0; u < 6; ++u) {                                        // unit = (channel block ks, row shift ty)
    const int ks = u / 3, ty = u % 3;
    if (u + 1 < 6) {
      const int ksn = (u + 1) / 3, tyn = (u + 1) % 3;
      const int o = gb + ksn * kstride + (tyn == 0 ? toff0 : (tyn == 1 ? toff1 : toff2));
      bn[0] = Xs[o]; bn[1] = Xs[PIECE + o];
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // term-major over the three column shifts: consecutive MFMAs go to different accumulators
      constexpr int TA[3] = {0, 1, 0}, TB[3] = {0, 0, 1};
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
          acc[ks][ty * 3 + tx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[tx][TA[t]], bc[TB[t]], acc[ks][ty * 3 + tx], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    bc[0] = bn[0]; bc[1] = bn[1];
  }
}

// the scale bookkeeping shared by both kernels: lower `s` to fit magnitude m, carrying the accumulators over (uniform branch)
__device__ __forceinline__ void hw_rescale(f32x4 (&acc)[2][9], float& s, float m) {
  const float sn = fminf(s, h2_scale_for(m));
  if (sn != s) {
    const float f = sn * h2_inv_pow2(s);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[ks][t] *= f;
    s = sn;
  }
}

template <int S, int BN, int BK>
__global__ __launch_bounds__(512, 1) void wgrad_h2(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                    int B, int K, int N, int tiles_per_split, int ntiles) {
  using G = HwGeo<S, BN, BK>;
  constexpr int HW = S * S, G8 = G::G8, NG = G::NG, NE = G::NE, PIECE = G::PIECE, NSG = G::NSG;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  h8* Xs = reinterpret_cast<h8*>(smem_raw);                           // [piece 2][ksub BK / 16][NG][16] records
  float* wmax = reinterpret_cast<float*>(smem_raw + (size_t)2 * PIECE * 16);   // the eight waves' tile maxima
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kgl = lane >> 4;
  const int role = wv % G::NR, sg = wv / G::NR;                       // wave group sg takes steps sg, sg + NSG, ..
  const int ns = role % (BN / 16), kh = role / (BN / 16);             // this wave: output channels n0 + 16 ns .., input channels k0 + 32 kh ..
  const int nkb = blockIdx.y, nbk = N / G::BN;
  const int n0 = (nkb % nbk) * G::BN, k0 = (nkb / nbk) * G::BK;
  const int split = blockIdx.x;
  const int tbeg = split * tiles_per_split, tend = min(ntiles, tbeg + tiles_per_split);

  // ---- X staging plan: task = one record (16-channel block, group, channel)
  int s_off[NE], s_rr[NE];                                            // element offset less the tile base; halo-inclusive row (-1: no task)
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int t = tid + 512 * e;
    s_rr[e] = -1; s_off[e] = 0;
    if (t < G::TASKS) {
      const int c16 = t & 15, g = (t >> 4) % NG, ksub = t / (16 * NG);
      const int i = g / G::IG, rem = g - i * G::IG, rr = rem / G8, c8 = rem - rr * G8;
      s_rr[e] = rr | (i << 8);
      s_off[e] = (i * K + ksub * 16 + c16) * HW + (rr - 1) * S + c8 * 8;
    }
  }
  float xr[NE][8];
  auto x_fetch = [&](int tile) {
    const int img0 = G::IPT > 1 ? tile * G::IPT : tile / G::TPI;
    const int row0 = G::IPT > 1 ? 0 : (tile % G::TPI) * G::R;
    const float* xb = x + ((long)img0 * K + k0) * HW + row0 * S;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int rr = s_rr[e] & 255, i = s_rr[e] >> 8, yy = row0 + rr - 1;
      const bool ok = s_rr[e] >= 0 && yy >= 0 && yy < S && img0 + i < B;
      const float4* p = reinterpret_cast<const float4*>(xb + (ok ? s_off[e] : 0));
      const float4 a = ok ? p[0] : make_float4(0.f, 0.f, 0.f, 0.f), b = ok ? p[1] : make_float4(0.f, 0.f, 0.f, 0.f);
      xr[e][0] = a.x; xr[e][1] = a.y; xr[e][2] = a.z; xr[e][3] = a.w; xr[e][4] = b.x; xr[e][5] = b.y; xr[e][6] = b.z; xr[e][7] = b.w;
    }
  };
  auto x_amax = [&]() {                                                // this wave's share of the staged tile's max |x| -> wmax[wv]
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(xr[e][j]));
    m = wave_amax(m);
    if (lane == 0) wmax[wv] = m;
  };
  auto x_commit = [&](float s) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (s_rr[e] < 0) continue;
      h8 p0, p1;
      h2_split8(xr[e], s, p0, p1);
      const int t = tid + 512 * e;
      Xs[t] = p0; Xs[PIECE + t] = p1;
    }
  };

  // ---- dY: lane (row n = l & 15, pixel group l >> 4 of the 32-pixel step) loads dy[p0 - 1 .. p0 + 8]
  float dv[10];
  const float* dyw = dy + (long)(n0 + 16 * ns + l15) * HW;
  auto d_fetch = [&](int tile, int step) {
    const int img0 = G::IPT > 1 ? tile * G::IPT : tile / G::TPI;
    const int row0 = G::IPT > 1 ? 0 : (tile % G::TPI) * G::R;
    const int pg = 4 * step + kgl;                                     // 8-pixel group of the tile
    const int i = pg / (G::R * G8), rem = pg - i * (G::R * G8), r = rem / G8, c8 = rem - r * G8;
    const bool ok = img0 + i < B;
    const float* p = dyw + (long)(img0 + i) * N * HW + (row0 + r) * S + c8 * 8;
    const float4 a = ok ? reinterpret_cast<const float4*>(p)[0] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b = ok ? reinterpret_cast<const float4*>(p)[1] : make_float4(0.f, 0.f, 0.f, 0.f);
    dv[1] = a.x; dv[2] = a.y; dv[3] = a.z; dv[4] = a.w; dv[5] = b.x; dv[6] = b.y; dv[7] = b.z; dv[8] = b.w;
    dv[0] = (G8 > 1 && ok && c8 > 0) ? p[-1] : 0.f;                    // the row's edge: dY outside its row counts as 0
    dv[9] = (G8 > 1 && ok && c8 < G8 - 1) ? p[8] : 0.f;
  };

  f32x4 acc[2][9];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[ks][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // B fragment of (16-channel block ks of this wave's half, row shift ty) at step `step`: record (piece, ksub, group, channel)
  const int brec0 = ((2 * kh) * NG) * 16 + l15;
  auto b_group = [&](int step) {                                       // halo-inclusive group index of this lane's 8 pixels, row shift 0
    const int pg = 4 * step + kgl;
    const int i = pg / (G::R * G8), rem = pg - i * (G::R * G8), r = rem / G8, c8 = rem - r * G8;
    return i * G::IG + (r + 1) * G8 + c8;
  };

  float sx = __uint_as_float(kH2ScaleCapBits), sd = sx;               // running scales: x (workgroup), dY (this wave)
  if (tbeg < tend) { x_fetch(tbeg); d_fetch(tbeg, sg); }
  for (int tile = tbeg; tile < tend; ++tile) {
    x_amax();
    __syncthreads();                                                   // the previous tile's fragment reads are done; the eight maxima are visible
    {
      float m = wmax[0];
#pragma unroll
      for (int q = 1; q < 8; ++q) m = fmaxf(m, wmax[q]);
      hw_rescale(acc, sx, m);
    }
    x_commit(sx);
    __syncthreads();
    if (tile + 1 < tend) x_fetch(tile + 1);                            // in flight during the multiplies
#pragma unroll 1
    for (int step = sg; step < 4; step += NSG) {
      // ---- A fragments: scale and split the 10 dY values once, pack the three column shifts (tx = 0, 1, 2 <-> dY[q+1], dY[q], dY[q-1])
      {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 10; ++j) m = fmaxf(m, fabsf(dv[j]));
        hw_rescale(acc, sd, wave_amax(m));
      }
      _Float16 pc[2][10];
#pragma unroll
      for (int j = 0; j < 10; ++j) h2_split(dv[j], sd, pc[0][j], pc[1][j]);
      h8 af[3][2];                                                     // [tx][piece]
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        u32x4 e0, e1, e2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          e0[q] = h_pack(pc[p][2 * q + 2], pc[p][2 * q + 3]);          // tx = 0: v[j + 2]
          e1[q] = h_pack(pc[p][2 * q + 1], pc[p][2 * q + 2]);          // tx = 1: v[j + 1]
          e2[q] = h_pack(pc[p][2 * q], pc[p][2 * q + 1]);              // tx = 2: v[j]
        }
        af[0][p] = __builtin_bit_cast(h8, e0); af[1][p] = __builtin_bit_cast(h8, e1); af[2][p] = __builtin_bit_cast(h8, e2);
      }
      // the next step's dY (or the next tile's first) in flight during the multiplies
      if (step + NSG < 4) d_fetch(tile, step + NSG); else if (tile + 1 < tend) d_fetch(tile + 1, sg);

      hw_step(acc, af, Xs, PIECE, brec0 + b_group(step) * 16, NG * 16, -G8 * 16, 0, G8 * 16);
    }
  }

  hw_finish<G::NR, NSG>(acc, h2_inv_pow2(sx), h2_inv_pow2(sd), smem_raw, part, split, role, sg, lane, n0 + 16 * ns, k0 + 32 * kh, N, K);
}


// ---- 4 x 4 maps -------------------------------------------------------------------------------------------------------------
// An 8-pixel group is TWO rows of an image, so a row shift of one is half a group: every (image, channel) keeps FIVE
// records per piece -- the aligned pairs (rows 0-1, 2-3: row shift 0) and the odd pairs (rows -1-0, 1-2, 3-4 with the
// outside rows zero: shifts -1 / +1).  A 128-pixel tile = 8 whole images, a 32-pixel step = 2 of them; the column shift
// stays inside each row of four (no neighbour loads).  One staging task = one (image, channel): 16 contiguous floats.
template <int BN, int BK>
__global__ __launch_bounds__(512, 1) void wgrad_h2_s4(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                       int B, int K, int N, int tiles_per_split, int ntiles) {
  constexpr int NR = (BN / 16) * (BK / 32), NSG = 8 / NR, NG = 40, PIECE = (BK / 16) * NG * 16, TASKS = 8 * BK;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  h8* Xs = reinterpret_cast<h8*>(smem_raw);                           // [piece 2][ksub BK / 16][image 8][record 5][16] records
  float* wmax = reinterpret_cast<float*>(smem_raw + (size_t)2 * PIECE * 16);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kgl = lane >> 4;
  const int role = wv % NR, sg = wv / NR;
  const int ns = role % (BN / 16), kh = role / (BN / 16);
  const int nkb = blockIdx.y, nbk = N / BN;
  const int n0 = (nkb % nbk) * BN, k0 = (nkb / nbk) * BK;
  const int split = blockIdx.x;
  const int tbeg = split * tiles_per_split, tend = min(ntiles, tbeg + tiles_per_split);

  const bool has_task = tid < TASKS;
  const int ti = tid / BK, tc = tid % BK;                              // staging task: image ti of the tile, channel tc
  float xr[16];
  auto x_fetch = [&](int tile) {
    const int img = tile * 8 + ti;
    const bool ok = has_task && img < B;
    const float4* p = reinterpret_cast<const float4*>(x + ((long)(ok ? img : 0) * K + k0 + tc) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = ok ? p[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      xr[4 * q] = v.x; xr[4 * q + 1] = v.y; xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
    }
  };
  auto x_amax = [&]() {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, fabsf(xr[j]));             // (threads without a task hold zeros)
    m = wave_amax(m);
    if (lane == 0) wmax[wv] = m;
  };
  auto x_commit = [&](float s) {
    if (!has_task) return;
    _Float16 pc[2][16];
#pragma unroll
    for (int j = 0; j < 16; ++j) h2_split(xr[j], s, pc[0][j], pc[1][j]);
    const _Float16 Z = (_Float16)0.f;
    const int rec = (((tc >> 4) * 8 + ti) * 5) * 16 + (tc & 15);       // record 0 of this (image, channel)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      h8 a0, a1, o0, o1, o2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0[j] = pc[p][j]; a1[j] = pc[p][8 + j]; o1[j] = pc[p][4 + j];
        o0[j] = j < 4 ? Z : pc[p][j - 4];                              // rows (-1, 0)
        o2[j] = j < 4 ? pc[p][12 + j] : Z;                             // rows (3, 4)
      }
      h8* d = Xs + p * PIECE + rec;
      d[0] = a0; d[16] = a1; d[32] = o0; d[48] = o1; d[64] = o2;
    }
  };

  float dv[8];
  const float* dyw = dy + (long)(n0 + 16 * ns + l15) * 16;
  auto d_fetch = [&](int tile, int step) {
    const int img = tile * 8 + 2 * step + (kgl >> 1);
    const bool ok = img < B;
    const float4* p = reinterpret_cast<const float4*>(dyw + (long)(ok ? img : 0) * N * 16 + (kgl & 1) * 8);
    const float4 a = ok ? p[0] : make_float4(0.f, 0.f, 0.f, 0.f), b = ok ? p[1] : make_float4(0.f, 0.f, 0.f, 0.f);
    dv[0] = a.x; dv[1] = a.y; dv[2] = a.z; dv[3] = a.w; dv[4] = b.x; dv[5] = b.y; dv[6] = b.z; dv[7] = b.w;
  };

  f32x4 acc[2][9];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[ks][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int brec0 = ((2 * kh) * NG) * 16 + l15;

  float sx = __uint_as_float(kH2ScaleCapBits), sd = sx;
  if (!has_task) {
#pragma unroll
    for (int j = 0; j < 16; ++j) xr[j] = 0.f;
  }
  if (tbeg < tend) { x_fetch(tbeg); d_fetch(tbeg, sg); }
  for (int tile = tbeg; tile < tend; ++tile) {
    x_amax();
    __syncthreads();
    {
      float m = wmax[0];
#pragma unroll
      for (int q = 1; q < 8; ++q) m = fmaxf(m, wmax[q]);
      hw_rescale(acc, sx, m);
    }
    x_commit(sx);
    __syncthreads();
    if (tile + 1 < tend) x_fetch(tile + 1);
#pragma unroll 1
    for (int step = sg; step < 4; step += NSG) {
      {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(dv[j]));
        hw_rescale(acc, sd, wave_amax(m));
      }
      _Float16 pc[2][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) h2_split(dv[j], sd, pc[0][j], pc[1][j]);
      const _Float16 Z = (_Float16)0.f;
      h8 af[3][2];                                                     // [tx][piece]: dY[q+1] | dY[q] | dY[q-1] inside each row of four
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          af[1][p][j] = pc[p][j];
          af[0][p][j] = (j & 3) == 3 ? Z : pc[p][(j + 1) & 7];
          af[2][p][j] = (j & 3) == 0 ? Z : pc[p][(j + 7) & 7];
        }
      if (step + NSG < 4) d_fetch(tile, step + NSG); else if (tile + 1 < tend) d_fetch(tile + 1, sg);
      // the lane's group: image 2 step + (kgl >> 1), row pair kgl & 1: aligned record (kgl & 1), odd records 2 + (kgl & 1) / 3 + (kgl & 1)
      const int gb = brec0 + ((2 * step + (kgl >> 1)) * 5 + (kgl & 1)) * 16;
      hw_step(acc, af, Xs, PIECE, gb, NG * 16, 2 * 16, 0, 3 * 16);
    }
  }
  hw_finish<NR, NSG>(acc, h2_inv_pow2(sx), h2_inv_pow2(sd), smem_raw, part, split, role, sg, lane, n0 + 16 * ns, k0 + 32 * kh, N, K);
}


// ------------------------------------------------------------------------------------------------------------------------
// 1x1 convolutions / token Linear layers:  dW[n][k] = sum_p dY[n][p] X[k][p],  db[n] = sum_p dY[n][p]   (as pw_wgrad_bf3)
// Both operands come straight from global memory; a wave scales each by its own running power of two (checked per 32-pixel
// step with ONE compare + ballot per operand: the full wave maximum is only taken when a step would overflow), splits into
// two fp16 pieces and issues 3 MFMAs per (16 x 16 x 32) block instead of 6.  The bias sums take the unscaled values.
// ------------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(512, 1) void pw_wgrad_h2(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                      float* __restrict__ bias_part, int B, int K, int N, int L, int steps_per_split,
                                                      int nsteps, int NR) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kgl = lane >> 4;
  const int WP = 8 / NR;                                               // pixel slices per workgroup
  const int role = blockIdx.y * NR + wv % NR, wp = wv / NR;
  const int ntn = N / (16 * NT);
  const int n0 = (role % ntn) * 16 * NT, k0 = (role / ntn) * 32;
  const int split = blockIdx.x;
  const int sbeg = split * steps_per_split, send = min(nsteps, sbeg + steps_per_split);

  f32x4 acc[NT][2];
  float bsum[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) { bsum[i] = 0.f; acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }

  float4 ra[NT][2], rb[2][2];
  auto fetch = [&](int t) {
    const long P0 = 32L * t + 8 * kgl;                                 // this lane's 8 pixels (L % 8 == 0: they never straddle two images)
    const long b = P0 / L, p = P0 - b * L;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const float4* q = reinterpret_cast<const float4*>(dy + (b * N + n0 + 16 * i + l15) * L + p);
      ra[i][0] = q[0]; ra[i][1] = q[1];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4* q = reinterpret_cast<const float4*>(x + (b * K + k0 + 16 * j + l15) * L + p);
      rb[j][0] = q[0]; rb[j][1] = q[1];
    }
  };
  auto amax8 = [&](const float4& u, const float4& v, float m) {
    m = fmaxf(fmaxf(m, fabsf(u.x)), fabsf(u.y)); m = fmaxf(fmaxf(m, fabsf(u.z)), fabsf(u.w));
    m = fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y)); m = fmaxf(fmaxf(m, fabsf(v.z)), fabsf(v.w));
    return m;
  };
  auto split8 = [&](const float4& u, const float4& v, float s, h8 (&f)[2]) {
    const float e[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    h2_split8(e, s, f[0], f[1]);
  };
  // lower running scale `s` if this lane-maximum m would leave fp16's range under it (uniform decision by ballot), carrying
  // the accumulators over; `lim` = 2^15 / s
  auto fit = [&](float& s, float& lim, float m) {
    if (__builtin_amdgcn_ballot_w64(!(m < lim)) != 0ull) {              // (rare; also taken for inf / nan)
      const float sn = fminf(s, h2_scale_for(wave_amax(m)));
      const float f = sn * h2_inv_pow2(s);
#pragma unroll
      for (int i = 0; i < NT; ++i) { acc[i][0] *= f; acc[i][1] *= f; }
      s = sn; lim = 32768.f * h2_inv_pow2(sn);
    }
  };
  float sa = __uint_as_float(kH2ScaleCapBits), sb = sa, lima = 0.f, limb = 0.f;   // lim = 0: the first step always sets the scales
  if (sbeg + wp < send) fetch(sbeg + wp);
#pragma unroll 1
  for (int t = sbeg + wp; t < send; t += WP) {
    float ma = 0.f, mb = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) ma = amax8(ra[i][0], ra[i][1], ma);
#pragma unroll
    for (int j = 0; j < 2; ++j) mb = amax8(rb[j][0], rb[j][1], mb);
    fit(sa, lima, ma);
    fit(sb, limb, mb);
    h8 fa[NT][2], fb[2][2];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      split8(ra[i][0], ra[i][1], sa, fa[i]);
      bsum[i] += ((ra[i][0].x + ra[i][0].y) + (ra[i][0].z + ra[i][0].w)) + ((ra[i][1].x + ra[i][1].y) + (ra[i][1].z + ra[i][1].w));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) split8(rb[j][0], rb[j][1], sb, fb[j]);
    if (t + WP < send) fetch(t + WP);                                  // in flight during the multiplies
    constexpr int TA[3] = {0, 1, 0}, TB[3] = {0, 0, 1};
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][TA[m]], fb[j][TB[m]], acc[i][j], 0, 0, 0);
  }
  {                                                                    // the true sums (every wave had its own scales)
    const float ua = h2_inv_pow2(sa), ub = h2_inv_pow2(sb);
#pragma unroll
    for (int i = 0; i < NT; ++i) { acc[i][0] = (acc[i][0] * ua) * ub; acc[i][1] = (acc[i][1] * ua) * ub; }
  }
  // bias: the four pixel groups of a row sit in lanes l15, l15 + 16, + 32, + 48
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    bsum[i] += __shfl_xor(bsum[i], 16, kWave);
    bsum[i] += __shfl_xor(bsum[i], 32, kWave);
  }
  // pixel slices 1 .. WP-1 hand their sums to slice 0 through LDS, in a fixed order
  float* red = reinterpret_cast<float*>(smem_raw);                    // [role in workgroup][NT * 8 + NT][64]
  const int rl = wv % NR;
#pragma unroll 1
  for (int g = 1; g < WP; ++g) {
    __syncthreads();
    if (wp == g) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) red[(rl * (NT * 9) + (i * 2 + j) * 4 + rg) * 64 + lane] = acc[i][j][rg];
        red[(rl * (NT * 9) + NT * 8 + i) * 64 + lane] = bsum[i];
      }
    }
    __syncthreads();
    if (wp == 0) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) acc[i][j][rg] += red[(rl * (NT * 9) + (i * 2 + j) * 4 + rg) * 64 + lane];
        bsum[i] += red[(rl * (NT * 9) + NT * 8 + i) * 64 + lane];
      }
    }
  }
  if (wp != 0) return;
  float* ps = part + (long)split * N * K;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) ps[(long)(n0 + 16 * i + 4 * kgl + rg) * K + k0 + 16 * j + l15] = acc[i][j][rg];
  if (bias_part && k0 == 0 && kgl == 0) {
#pragma unroll
    for (int i = 0; i < NT; ++i) bias_part[(long)split * N + n0 + 16 * i + l15] = bsum[i];
  }
}

// ---- host side (the plan -- tiles per split, block sizes -- is bf3_wgrad.hip's) ---------------------------------------------
int wgrad_bf3_plan(int B, int Cin, int Cout, int H, int W, int* tps, int* ntiles);
void wgrad_bf3_tile_for(int Cin, int Cout, long nt, int* bn, int* bk);

template <int S, int BN, int BK>
static void wgrad_h2_launch_t(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int tps, int nt, int splits,
                              hipStream_t s) {
  using G = HwGeo<S, BN, BK>;
  const size_t lds = std::max((size_t)2 * G::PIECE * 16 + 32, G::NSG > 1 ? sizeof(float) * G::NR * 72 * 64 : (size_t)0);
  (void)lds_opt_in(&wgrad_h2<S, BN, BK>, lds);                    // (> 64 KB of dynamic LDS: once per device and kernel)
  hipLaunchKernelGGL((wgrad_h2<S, BN, BK>), dim3((unsigned)splits, (unsigned)((Cout / BN) * (Cin / BK))), dim3(512), lds, s, x, dy,
                     part, B, Cin, Cout, tps, nt);
}

template <int BN, int BK>
static void wgrad_h2_s4_launch_t(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int tps, int nt, int splits,
                                 hipStream_t s) {
  constexpr int NR = (BN / 16) * (BK / 32);
  const size_t lds = std::max((size_t)2 * (BK / 16) * 40 * 16 * 16 + 32, 8 / NR > 1 ? sizeof(float) * NR * 72 * 64 : (size_t)0);
  (void)lds_opt_in(&wgrad_h2_s4<BN, BK>, lds);                    // (> 64 KB of dynamic LDS: once per device and kernel)
  hipLaunchKernelGGL((wgrad_h2_s4<BN, BK>), dim3((unsigned)splits, (unsigned)((Cout / BN) * (Cin / BK))), dim3(512), lds, s, x, dy, part, B,
                     Cin, Cout, tps, nt);
}

// writes `slabs` partial [9][Cout][Cin] slabs into part; the caller folds them
int wgrad_h2(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int H, int W, hipStream_t s) {
  int tps, nt;
  const int splits = wgrad_bf3_plan(B, Cin, Cout, H, W, &tps, &nt);
  if (!splits) return 0;
  int bn_, bk_;
  wgrad_bf3_tile_for(Cin, Cout, nt, &bn_, &bk_);
  const bool n64 = bn_ == 64, k64 = bk_ == 64;
#define AFD_WGH(S_)                                                                                      \
  if (n64 && k64) wgrad_h2_launch_t<S_, 64, 64>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);          \
  else if (n64) wgrad_h2_launch_t<S_, 64, 32>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);            \
  else if (k64) wgrad_h2_launch_t<S_, 32, 64>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);            \
  else wgrad_h2_launch_t<S_, 32, 32>(x, dy, part, B, Cin, Cout, tps, nt, splits, s)
  if (W == 32) { AFD_WGH(32); } else if (W == 16) { AFD_WGH(16); } else if (W == 8) { AFD_WGH(8); }
  else if (n64 && k64) wgrad_h2_s4_launch_t<64, 64>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);
  else if (n64) wgrad_h2_s4_launch_t<64, 32>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);
  else if (k64) wgrad_h2_s4_launch_t<32, 64>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);
  else wgrad_h2_s4_launch_t<32, 32>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);
#undef AFD_WGH
  return splits;
}


int pw_wgrad_bf3_plan(int B, int Cin, int Cout, int L, int* sps, int* nsteps, int* nr, int* nt);
int pw_wgrad_h2(const float* x, const float* dy, float* part, float* bias_part, int B, int Cin, int Cout, int L, hipStream_t s) {
  int sps, st, NR, NT;
  const int splits = pw_wgrad_bf3_plan(B, Cin, Cout, L, &sps, &st, &NR, &NT);
  if (!splits) return 0;
  const int roles = (Cout / (16 * NT)) * (Cin / 32);
  const size_t lds = 8 / NR > 1 ? sizeof(float) * NR * NT * 9 * 64 : 0;
  const dim3 grid((unsigned)splits, (unsigned)(roles / NR));
  if (NT == 3) hipLaunchKernelGGL(pw_wgrad_h2<3>, grid, dim3(512), lds, s, x, dy, part, bias_part, B, Cin, Cout, L, sps, st, NR);
  else hipLaunchKernelGGL(pw_wgrad_h2<2>, grid, dim3(512), lds, s, x, dy, part, bias_part, B, Cin, Cout, L, sps, st, NR);
  return splits;
}

}  // namespace afd

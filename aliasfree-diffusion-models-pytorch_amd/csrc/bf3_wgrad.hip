// bf3_wgrad.hip -- F5: weight gradient of the 3x3 (pad 1) convolution on the bf16 matrix cores at fp32 accuracy (round 2).
//
//   dW[n][k][ty][tx] = sum over (image, pixel p) of dY[n][p] * X[k][p + (ty-1) W + (tx-1)]
//
// is a GEMM whose REDUCTION runs over pixels.  Both operands split exactly into three bf16 pieces (bf3_weights.h) and the
// six leading cross terms go to v_mfma_f32_16x16x32_bf16 -- 6/16 of an fp32-MFMA per multiply instead of Winograd's
// 16/36, on the matrix pipe instead of the vector pipe (wgrad_wino of wino.hip is bound by the fp32 MFMA issue rate).
//
// The MFMA wants, per lane, 8 CONSECUTIVE reduction elements of one row / column = 8 consecutive pixels of one channel.
// A tap shifts X by (ty-1) rows and (tx-1) columns.  The row shift is a whole number of 8-pixel groups (W = 8, 16, 32),
// the column shift is not -- so the column shift moves to the dY side (re-index q = p + tx - 1):
//
//   dW[n][k][ty][tx] = sum_q dY[n][q - (tx-1)] * X[k][q + (ty-1) W],       dY taken as 0 outside its row
//
//   B operand (X):  staged once per 128-pixel tile, split, kept in LDS as 16-byte records [piece][16-channel block][group
//                   of 8 pixels, halo rows zero][channel]; the fragment of lane (channel l & 15, group l >> 4) for row
//                   shift ty is ONE ds_read_b128 (consecutive lanes = consecutive records: conflict-free).
//   A operand (dY): each wave reads ITS 16 output channels straight from global memory (32 contiguous bytes per lane +
//                   the two neighbours), splits the 10 values once and packs the three column-shifted fragments from them
//                   in registers.
//   A workgroup (8 waves) owns BN x BK = 64 / 32 output x 64 / 32 input channels and a range of pixel tiles; a wave role
//   (ns, kh) accumulates 16 n x 32 k x 9 taps (72 accumulator registers).  With fewer than 8 roles (32-channel layers)
//   the 8 / roles wave groups take the 32-pixel steps of a tile in turn and are summed through LDS at the end.  Partial slabs
//   [split][tap][cout][cin] + the deterministic wgrad_reduce of conv.hip, as for the other wgrad forms.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include "common.h"
#include "bf3_weights.h"

namespace afd {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

template <int S, int BN_, int BK_> struct BwGeo {
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

__device__ __forceinline__ uint32_t bf_pack(__bf16 lo, __bf16 hi) {
  return (uint32_t)__builtin_bit_cast(uint16_t, lo) | ((uint32_t)__builtin_bit_cast(uint16_t, hi) << 16);
}

// wave groups 1 .. NSG-1 hand their sums to group 0 through LDS, in a fixed order; group 0 writes the slab [split][tap][cout][cin]
// (accumulator: row n = 4 (l >> 4) + reg, column k = l & 15)
template <int NR, int NSG>
__device__ __forceinline__ void wg_finish(f32x4 (&acc)[2][9], uint8_t* smem_raw, float* __restrict__ part, int split, int role, int sg,
                                          int lane, int n0w, int k0w, int N, int K) {
  const int l15 = lane & 15, kgl = lane >> 4;
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

// the 54 + 54 MFMAs of one 32-pixel step: units (16-channel block ks, row shift ty), B fragments one unit ahead; gb = record of
// (piece 0, this wave's first channel block, the lane's group at row shift 0, channel l & 15), kstride = records per channel
// block, toff[ty] = record offset of row shift ty
__device__ __forceinline__ void wg_step(f32x4 (&acc)[2][9], const bf8 (&af)[3][3], const bf8* __restrict__ Xs, int PIECE, int gb, int kstride,
                                        int toff0, int toff1, int toff2) {
  bf8 bc[3], bn[3];
  { const int o = gb + toff0; bc[0] = Xs[o]; bc[1] = Xs[PIECE + o]; bc[2] = Xs[2 * PIECE + o]; }
#pragma unroll
  for (int u = 0; u < 6; ++u) {                                        // unit = (channel block ks, row shift ty)
    const int ks = u / 3, ty = u % 3;
    if (u + 1 < 6) {
      const int ksn = (u + 1) / 3, tyn = (u + 1) % 3;
      const int o = gb + ksn * kstride + (tyn == 0 ? toff0 : (tyn == 1 ? toff1 : toff2));
      bn[0] = Xs[o]; bn[1] = Xs[PIECE + o]; bn[2] = Xs[2 * PIECE + o];
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // term-major over the three column shifts: consecutive MFMAs go to different accumulators
      constexpr int TA[6] = {0, 1, 0, 2, 1, 0}, TB[6] = {0, 0, 1, 0, 1, 2};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
          acc[ks][ty * 3 + tx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tx][TA[t]], bc[TB[t]], acc[ks][ty * 3 + tx], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    bc[0] = bn[0]; bc[1] = bn[1]; bc[2] = bn[2];
  }
}

template <int S, int BN, int BK>
__global__ __launch_bounds__(512, 1) void wgrad_bf3(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                    int B, int K, int N, int tiles_per_split, int ntiles) {
  using G = BwGeo<S, BN, BK>;
  constexpr int HW = S * S, G8 = G::G8, NG = G::NG, NE = G::NE, PIECE = G::PIECE, NSG = G::NSG;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  bf8* Xs = reinterpret_cast<bf8*>(smem_raw);                         // [piece 3][ksub BK / 16][NG][16] records
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
  auto x_commit = [&]() {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (s_rr[e] < 0) continue;
      bf8 p0, p1, p2;
#pragma unroll
      for (int j = 0; j < 8; ++j) { __bf16 a, b, c; bf3_split(xr[e][j], a, b, c); p0[j] = a; p1[j] = b; p2[j] = c; }
      const int t = tid + 512 * e;
      Xs[t] = p0; Xs[PIECE + t] = p1; Xs[2 * PIECE + t] = p2;
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

  if (tbeg < tend) { x_fetch(tbeg); d_fetch(tbeg, sg); }
  for (int tile = tbeg; tile < tend; ++tile) {
    __syncthreads();                                                   // the previous tile's fragment reads are done
    x_commit();
    __syncthreads();
    if (tile + 1 < tend) x_fetch(tile + 1);                            // in flight during the multiplies
#pragma unroll 1
    for (int step = sg; step < 4; step += NSG) {
      // ---- A fragments: split the 10 dY values once, pack the three column shifts (tx = 0, 1, 2 <-> dY[q+1], dY[q], dY[q-1])
      __bf16 pc[3][10];
#pragma unroll
      for (int j = 0; j < 10; ++j) bf3_split(dv[j], pc[0][j], pc[1][j], pc[2][j]);
      bf8 af[3][3];                                                    // [tx][piece]
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        u32x4 e0, e1, e2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          e0[q] = bf_pack(pc[p][2 * q + 2], pc[p][2 * q + 3]);         // tx = 0: v[j + 2]
          e1[q] = bf_pack(pc[p][2 * q + 1], pc[p][2 * q + 2]);         // tx = 1: v[j + 1]
          e2[q] = bf_pack(pc[p][2 * q], pc[p][2 * q + 1]);             // tx = 2: v[j]
        }
        af[0][p] = __builtin_bit_cast(bf8, e0); af[1][p] = __builtin_bit_cast(bf8, e1); af[2][p] = __builtin_bit_cast(bf8, e2);
      }
      // the next step's dY (or the next tile's first) in flight during the multiplies
      if (step + NSG < 4) d_fetch(tile, step + NSG); else if (tile + 1 < tend) d_fetch(tile + 1, sg);

      wg_step(acc, af, Xs, PIECE, brec0 + b_group(step) * 16, NG * 16, -G8 * 16, 0, G8 * 16);
    }
  }

  wg_finish<G::NR, NSG>(acc, smem_raw, part, split, role, sg, lane, n0 + 16 * ns, k0 + 32 * kh, N, K);
}


// ---- 4 x 4 maps -------------------------------------------------------------------------------------------------------------
// An 8-pixel group is TWO rows of an image, so a row shift of one is half a group: every (image, channel) keeps FIVE
// records per piece -- the aligned pairs (rows 0-1, 2-3: row shift 0) and the odd pairs (rows -1-0, 1-2, 3-4 with the
// outside rows zero: shifts -1 / +1).  A 128-pixel tile = 8 whole images, a 32-pixel step = 2 of them; the column shift
// stays inside each row of four (no neighbour loads).  One staging task = one (image, channel): 16 contiguous floats.
template <int BN, int BK>
__global__ __launch_bounds__(512, 1) void wgrad_bf3_s4(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                       int B, int K, int N, int tiles_per_split, int ntiles) {
  constexpr int NR = (BN / 16) * (BK / 32), NSG = 8 / NR, NG = 40, PIECE = (BK / 16) * NG * 16, TASKS = 8 * BK;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  bf8* Xs = reinterpret_cast<bf8*>(smem_raw);                         // [piece 3][ksub BK / 16][image 8][record 5][16] records
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
  auto x_commit = [&]() {
    if (!has_task) return;
    __bf16 pc[3][16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bf3_split(xr[j], pc[0][j], pc[1][j], pc[2][j]);
    const __bf16 Z = (__bf16)0.f;
    const int rec = (((tc >> 4) * 8 + ti) * 5) * 16 + (tc & 15);       // record 0 of this (image, channel)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      bf8 a0, a1, o0, o1, o2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0[j] = pc[p][j]; a1[j] = pc[p][8 + j]; o1[j] = pc[p][4 + j];
        o0[j] = j < 4 ? Z : pc[p][j - 4];                              // rows (-1, 0)
        o2[j] = j < 4 ? pc[p][12 + j] : Z;                             // rows (3, 4)
      }
      bf8* d = Xs + p * PIECE + rec;
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

  if (tbeg < tend) { x_fetch(tbeg); d_fetch(tbeg, sg); }
  for (int tile = tbeg; tile < tend; ++tile) {
    __syncthreads();
    x_commit();
    __syncthreads();
    if (tile + 1 < tend) x_fetch(tile + 1);
#pragma unroll 1
    for (int step = sg; step < 4; step += NSG) {
      __bf16 pc[3][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bf3_split(dv[j], pc[0][j], pc[1][j], pc[2][j]);
      const __bf16 Z = (__bf16)0.f;
      bf8 af[3][3];                                                    // [tx][piece]: dY[q+1] | dY[q] | dY[q-1] inside each row of four
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          af[1][p][j] = pc[p][j];
          af[0][p][j] = (j & 3) == 3 ? Z : pc[p][(j + 1) & 7];
          af[2][p][j] = (j & 3) == 0 ? Z : pc[p][(j + 7) & 7];
        }
      if (step + NSG < 4) d_fetch(tile, step + NSG); else if (tile + 1 < tend) d_fetch(tile + 1, sg);
      // the lane's group: image 2 step + (kgl >> 1), row pair kgl & 1: aligned record (kgl & 1), odd records 2 + (kgl & 1) / 3 + (kgl & 1)
      const int gb = brec0 + ((2 * step + (kgl >> 1)) * 5 + (kgl & 1)) * 16;
      wg_step(acc, af, Xs, PIECE, gb, NG * 16, 2 * 16, 0, 3 * 16);
    }
  }
  wg_finish<NR, NSG>(acc, smem_raw, part, split, role, sg, lane, n0 + 16 * ns, k0 + 32 * kh, N, K);
}

// ------------------------------------------------------------------------------------------------------------------------
// 1x1 convolutions / token Linear layers:  dW[n][k] = sum_p dY[n][p] X[k][p],  db[n] = sum_p dY[n][p]
// No shifts, so BOTH operands come straight from global memory: lane (channel l & 15, pixel group l >> 4) loads 32
// contiguous bytes, splits them and holds the fragments of a 32-pixel step.  A wave owns 16 NT output x 32 input channels;
// the 8 waves of a workgroup are (output tile, input tile) roles x pixel slices, summed through LDS at the end; the pixel
// range is split over workgroups (slabs [split][cout][cin] + bias slabs, reduced by wgrad_reduce).  Each operand element
// is read once per role that needs it (L2) and once from HBM: the kernel is HBM-bound.
// ------------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(512, 1) void pw_wgrad_bf3(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
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
  auto split8 = [&](const float4& u, const float4& v, bf8 (&f)[3]) {
    const float e[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) { __bf16 a, b, c; bf3_split(e[j], a, b, c); f[0][j] = a; f[1][j] = b; f[2][j] = c; }
  };
  if (sbeg + wp < send) fetch(sbeg + wp);
#pragma unroll 1
  for (int t = sbeg + wp; t < send; t += WP) {
    bf8 fa[NT][3], fb[2][3];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      split8(ra[i][0], ra[i][1], fa[i]);
      bsum[i] += ((ra[i][0].x + ra[i][0].y) + (ra[i][0].z + ra[i][0].w)) + ((ra[i][1].x + ra[i][1].y) + (ra[i][1].z + ra[i][1].w));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) split8(rb[j][0], rb[j][1], fb[j]);
    if (t + WP < send) fetch(t + WP);                                  // in flight during the multiplies
    constexpr int TA[6] = {0, 1, 0, 2, 1, 0}, TB[6] = {0, 0, 1, 0, 1, 2};
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][TA[m]], fb[j][TB[m]], acc[i][j], 0, 0, 0);
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


// ------------------------------------------------------------------------------------------------------------------------
// The network's first layer (3 input channels): dW[n][c][ty][tx] is 27 numbers per output channel, the layer reads 12 B of
// x per 4 N B of dY -- memory-bound, and the LAST weight gradient of a step (nothing left to overlap with).  Vector FMAs:
// a workgroup (8 waves) takes whole images; the image sits zero-haloed in LDS, lane = pixel, wave w owns 4 output channels
// x 27 taps = 108 running sums per lane; wave-reduced once at the end, one slab per workgroup (fixed order, no atomics).
// ------------------------------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(512, 1) void wgrad_cin3(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                     int B, int N) {
  constexpr int RS = S + 3, PL = (S + 2) * RS;                         // LDS row stride, plane size (odd stride: rows spread over the banks)
  __shared__ float Xs[3 * PL];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n0 = blockIdx.y * 32 + 4 * wv;
  for (int i = tid; i < 3 * PL; i += 512) Xs[i] = 0.f;                 // the halo stays zero
  float acc[4][27];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[j][t] = 0.f;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();                                                   // the previous image's reads are done (first: the zero fill)
    for (int i = tid; i < 3 * S * S; i += 512) {
      const int c = i / (S * S), rem = i - c * S * S, r = rem / S, col = rem - r * S;
      Xs[c * PL + (r + 1) * RS + col + 1] = x[((long)b * 3 + c) * S * S + rem];
    }
    __syncthreads();
    const float* dyb = dy + ((long)b * N + n0) * S * S;
    float dn[4];                                                       // the next iteration's dY in flight during the FMAs
#pragma unroll
    for (int j = 0; j < 4; ++j) dn[j] = dyb[(long)j * S * S + lane];
#pragma unroll 1
    for (int p = lane; p < S * S; p += 64) {
      const int r = p / S, col = p - r * S;
      float d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = dn[j];
      if (p + 64 < S * S) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dn[j] = dyb[(long)j * S * S + p + 64];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            const float xv = Xs[c * PL + (r + ty) * RS + col + tx];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][(ty * 3 + tx) * 3 + c] = fmaf(d[j], xv, acc[j][(ty * 3 + tx) * 3 + c]);
          }
    }
  }
  float* ps = part + (long)blockIdx.x * 27 * N;                         // slab [tap][cout][cin = 3]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const float v = wave_sum_to_lane63(acc[j][t]);
      if (lane == 63) ps[((long)(t / 3) * N + n0 + j) * 3 + (t % 3)] = v;
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static int g_wgbf3_mode = 0;       // afd_debug_conv_path 84 / 85 / 86: by the rule / off / wherever the shape is covered
void wgrad_bf3_set_mode(int m) { g_wgbf3_mode = m; }

// the (output, input) channel block of a workgroup: 64 x 64 where the layer has >= 1024 (tile, block) work items -- twice
// the operand reuse per MFMA, 18 % faster on the large layers -- else 32 x 32: four times the blocks per split, so a quarter
// of the splits and of the slab bytes, 15-30 % faster on the 4x4 maps and the thin 8x8 layers (tools/wgrad_bench.py,
// AFD_WGB_TILE=32 / 64 forces one size)
static void wgrad_bf3_tile(int Cin, int Cout, long nt, int* bn, int* bk) {
  static const int force = [] { const char* e = getenv("AFD_WGB_TILE"); return e ? atoi(e) : 0; }();
  const bool can = Cout % 64 == 0 && Cin % 64 == 0;
  const bool big = force ? force >= 64 : nt * (Cout / 64) * (Cin / 64) >= 1024;
  if (can && big) { *bn = 64; *bk = 64; return; }
  *bn = (Cout % 64 == 0 && Cin % 64 && (force ? force >= 64 : true)) ? 64 : 32;   // mixed widths keep the 64-wide side
  *bk = (Cin % 64 == 0 && Cout % 64 && (force ? force >= 64 : true)) ? 64 : 32;
}

void wgrad_bf3_tile_for(int Cin, int Cout, long nt, int* bn, int* bk) { wgrad_bf3_tile(Cin, Cout, nt, bn, bk); }   // (h2_wgrad.hip shares the plan)
static int g_wg_arith_bf3 = 0;     // afd_debug_conv_path 78 / 79: the 3x3 weight gradient's arithmetic is f16x2 (h2_wgrad.hip, default) / bf16x3 (this file)
void wgrad_arith_set(int m) { g_wg_arith_bf3 = m; }
bool wgrad_arith_is_bf3() { return g_wg_arith_bf3 != 0; }

// plan: the number of slabs (0 = not covered / not chosen) and the tiles per split
static long g_wgb_target = 256;
void wgrad_plan_target_set(long wgs) { g_wgb_target = wgs; }

int wgrad_bf3_plan(int B, int Cin, int Cout, int H, int W, int* tps, int* ntiles) {
  if (g_wgbf3_mode == 1) return 0;
  if (H != W || (W != 4 && W != 8 && W != 16 && W != 32)) return 0;
  if (Cin % 32 || Cout % 32) return 0;
  if ((long)B * H * W * (Cin > Cout ? Cin : Cout) >= (1L << 31)) return 0;
  const long nt = W == 4 ? (B + 7) / 8 : (W == 8 ? (B + 1) / 2 : (long)B * (H * W / 128));
  int BN, BK;
  wgrad_bf3_tile(Cin, Cout, W == 4 ? (B + 7) / 8 : (W == 8 ? (B + 1) / 2 : (long)B * (H * W / 128)), &BN, &BK);
  const long blocks = (long)(Cout / BN) * (Cin / BK);
  // workgroups per launch (512 threads each, a CU's whole register file).  Alone a launch is fastest with one per CU (256: the default); IN SITU,
  // beside the dependent chain on the other stream, fewer are better -- they leave CUs to the chain and write fewer slabs:
  // step time with 64 / 96 / 128 / 160 / 192 / 256 / 384 / 512: 8.00 / 7.44 / 7.06 / 7.07 / 7.08 / 7.14 / 7.24 / 7.42 ms (tools/step_median.py --lanes)
  static const long env_target = [] { const char* e = getenv("AFD_WGB_TARGET"); return e ? atol(e) : 0L; }();   // tuning hook
  const long target = env_target > 0 ? env_target : g_wgb_target;       // 256 standalone, 160 beside the chain (afd_debug_conv_path 48 / 49)
  long s = target / blocks;
  if (s < 1) s = 1;
  if (s > nt) s = nt;
  const long c = (nt + s - 1) / s;
  if (g_wgbf3_mode != 2 && nt * blocks < 128) return 0;               // too little work to fill half the chip: the other forms
  *tps = (int)c; *ntiles = (int)nt;
  return (int)((nt + c - 1) / c);
}

template <int S, int BN, int BK>
static void wgrad_bf3_launch_t(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int tps, int nt, int splits,
                               hipStream_t s) {
  using G = BwGeo<S, BN, BK>;
  const size_t lds = std::max((size_t)3 * G::PIECE * 16, G::NSG > 1 ? sizeof(float) * G::NR * 72 * 64 : (size_t)0);
  (void)lds_opt_in(&wgrad_bf3<S, BN, BK>, lds);                    // (> 64 KB of dynamic LDS: once per device and kernel)
  hipLaunchKernelGGL((wgrad_bf3<S, BN, BK>), dim3((unsigned)splits, (unsigned)((Cout / BN) * (Cin / BK))), dim3(512), lds, s, x, dy,
                     part, B, Cin, Cout, tps, nt);
}

template <int BN, int BK>
static void wgrad_bf3_s4_launch_t(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int tps, int nt, int splits,
                                  hipStream_t s) {
  constexpr int NR = (BN / 16) * (BK / 32);
  const size_t lds = std::max((size_t)3 * (BK / 16) * 40 * 16 * 16, 8 / NR > 1 ? sizeof(float) * NR * 72 * 64 : (size_t)0);
  (void)lds_opt_in(&wgrad_bf3_s4<BN, BK>, lds);                    // (> 64 KB of dynamic LDS: once per device and kernel)
  hipLaunchKernelGGL((wgrad_bf3_s4<BN, BK>), dim3((unsigned)splits, (unsigned)((Cout / BN) * (Cin / BK))), dim3(512), lds, s, x, dy, part, B,
                     Cin, Cout, tps, nt);
}

// writes `slabs` partial [9][Cout][Cin] slabs into part; the caller reduces them (wgrad_reduce)
int wgrad_bf3(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int H, int W, hipStream_t s) {
  int tps, nt;
  const int splits = wgrad_bf3_plan(B, Cin, Cout, H, W, &tps, &nt);
  if (!splits) return 0;
  int bn_, bk_;
  wgrad_bf3_tile(Cin, Cout, nt, &bn_, &bk_);
  const bool n64 = bn_ == 64, k64 = bk_ == 64;
#define AFD_WGB(S_)                                                                                       \
  if (n64 && k64) wgrad_bf3_launch_t<S_, 64, 64>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);          \
  else if (n64) wgrad_bf3_launch_t<S_, 64, 32>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);            \
  else if (k64) wgrad_bf3_launch_t<S_, 32, 64>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);            \
  else wgrad_bf3_launch_t<S_, 32, 32>(x, dy, part, B, Cin, Cout, tps, nt, splits, s)
  if (W == 32) { AFD_WGB(32); } else if (W == 16) { AFD_WGB(16); } else if (W == 8) { AFD_WGB(8); }
  else if (n64 && k64) wgrad_bf3_s4_launch_t<64, 64>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);
  else if (n64) wgrad_bf3_s4_launch_t<64, 32>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);
  else if (k64) wgrad_bf3_s4_launch_t<32, 64>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);
  else wgrad_bf3_s4_launch_t<32, 32>(x, dy, part, B, Cin, Cout, tps, nt, splits, s);
#undef AFD_WGB
  return splits;
}


// 1x1 plan: slabs (0 = not covered / not chosen); fills the steps per split, the roles per workgroup and NT
static int g_pwbf3_mode = 0;       // afd_debug_conv_path 88 / 89: the 1x1 form by the rule / off
void pw_wgrad_bf3_set_mode(int m) { g_pwbf3_mode = m; }
int pw_wgrad_bf3_plan(int B, int Cin, int Cout, int L, int* sps, int* nsteps, int* nr, int* nt) {
  if (g_wgbf3_mode == 1 || g_pwbf3_mode == 1) return 0;
  if (Cin % 32 || Cout % 32 || L % 8 || ((long)B * L) % 32) return 0;    // (L = 16: a 32-pixel step takes two 4x4 images)
  if ((long)B * L * (Cin > Cout ? Cin : Cout) >= (1L << 31)) return 0;
  const int NT = Cout % 48 == 0 ? 3 : 2;
  const int roles = (Cout / (16 * NT)) * (Cin / 32);
  int NR = 8;
  while (roles % NR) NR >>= 1;                                         // roles per workgroup: a power of two dividing the role count
  const long st = (long)B * L / 32;
  const long gy = roles / NR;
  const char* e = getenv("AFD_PWB_TARGET");                            // tuning hooks, read per call (tools/ab_env.py)
  const long target = e ? atol(e) : 256L;
  const char* e2 = getenv("AFD_PWB_MINSTEPS");
  const long minsteps = (e2 ? atol(e2) : 1L) * (8 / NR);               // steps per split: at least this many per pixel slice of a workgroup
  long s = target / gy;
  if (s > st / minsteps) s = st / minsteps;
  if (s < 1) s = 1;
  if (s > st) s = st;
  const long c = (st + s - 1) / s;
  *sps = (int)c; *nsteps = (int)st; *nr = NR; *nt = NT;
  return (int)((st + c - 1) / c);
}

int pw_wgrad_bf3(const float* x, const float* dy, float* part, float* bias_part, int B, int Cin, int Cout, int L, hipStream_t s) {
  int sps, st, NR, NT;
  const int splits = pw_wgrad_bf3_plan(B, Cin, Cout, L, &sps, &st, &NR, &NT);
  if (!splits) return 0;
  const int roles = (Cout / (16 * NT)) * (Cin / 32);
  const size_t lds = 8 / NR > 1 ? sizeof(float) * NR * NT * 9 * 64 : 0;
  const dim3 grid((unsigned)splits, (unsigned)(roles / NR));
  if (NT == 3) hipLaunchKernelGGL(pw_wgrad_bf3<3>, grid, dim3(512), lds, s, x, dy, part, bias_part, B, Cin, Cout, L, sps, st, NR);
  else hipLaunchKernelGGL(pw_wgrad_bf3<2>, grid, dim3(512), lds, s, x, dy, part, bias_part, B, Cin, Cout, L, sps, st, NR);
  return splits;
}


// first-layer form: slabs (0 = not covered)
int wgrad_cin3_plan(int B, int Cin, int Cout, int H, int W) {
  if (g_wgbf3_mode == 1) return 0;
  if (Cin != 3 || Cout % 32 || H != W || (W != 16 && W != 32 && W != 64)) return 0;
  return B < 256 ? B : 256;
}
int wgrad_cin3(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int H, int W, hipStream_t s) {
  const int slabs = wgrad_cin3_plan(B, Cin, Cout, H, W);
  if (!slabs) return 0;
  const dim3 grid((unsigned)slabs, (unsigned)(Cout / 32));
  if (W == 64) hipLaunchKernelGGL(wgrad_cin3<64>, grid, dim3(512), 0, s, x, dy, part, B, Cout);
  else if (W == 32) hipLaunchKernelGGL(wgrad_cin3<32>, grid, dim3(512), 0, s, x, dy, part, B, Cout);
  else hipLaunchKernelGGL(wgrad_cin3<16>, grid, dim3(512), 0, s, x, dy, part, B, Cout);
  return slabs;
}

}  // namespace afd

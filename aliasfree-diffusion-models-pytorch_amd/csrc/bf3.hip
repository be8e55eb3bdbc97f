// bf3.hip -- F5: 3x3 (pad 1) convolution forward / dgrad as a DIRECT implicit GEMM on the bf16 matrix cores at fp32
// accuracy (round 2).
//
// Why: on gfx950 the fp32-input MFMA runs at the fp32 VECTOR rate and shares the vector pipe, so the Winograd kernels of
// wino.hip (16 fp32 multiplies per tile and channel pair, transforms on the same pipe) are at 0.94 of the fp32 matrix
// peak counting algorithmic flops and cannot go further.  The bf16 MFMA is 16x faster per instruction and has its own
// pipe.  Every fp32 value splits EXACTLY into three bf16 pieces (x = x1 + x2 + x3, 3 x 8 mantissa bits); the six leading
// cross terms of a product (a1 b1, a2 b1, a1 b2, a3 b1, a2 b2, a1 b3) carry it to 2^-24, every piece product is exact in
// the fp32 accumulator.  A direct 3x3 convolution then costs 36 x 6 / 16 = 13.5 fp32-MFMA-equivalents per output and
// channel pair -- less than Winograd's 16 -- with NO transforms: the weights are split once per step, the input once
// per staging (one element feeds 9 taps x all output channels of the workgroup), and the splitting (vector pipe) overlaps
// the multiplies (matrix pipe).  Measured max error 1.6e-7 of sum|a b| on this scheme (attention kernels, round 1).
//
//   bf3_weights    Wp[piece][tap][k/8][n][8 bf16]: 16-byte records = the A fragment of one lane
//                  (v_mfma_f32_16x16x32_bf16: lane (row n = l & 15, group l >> 4) holds k = 8 (l >> 4) + j); forward:
//                  n = cout, k = cin; dgrad: n = cin, k = cout, taps rotated by 180 degrees -- ONE main kernel serves
//                  both passes.  (16x16x32, not 32x32x16: a dense bf16 MFMA loop is power-limited on this chip and the
//                  smaller shape holds the higher clock -- 96 -> 77 us on 128->128 @16x16 from the shape change alone.)
//   conv_bf3       a workgroup owns NBLK x 32 output channels x 128 output pixels (whole rows of one image, or whole
//                  images).  Per chunk of 32 input channels the haloed input pixels go global -> registers -> split ->
//                  LDS as Xp[piece][k/8][pixel][8 bf16] (the next chunk's loads in flight during the multiplies); a
//                  B fragment is one ds_read_b128 at (pixel + tap offset), consecutive lanes = consecutive pixels:
//                  conflict-free.  A fragments come straight from global memory / L2 (consecutive lanes = consecutive
//                  16-byte records), one or two (tap, chunk) groups ahead.  6 MFMAs per (tap, chunk, 16 x 16 block).
#include <algorithm>
#include <cstdint>
#include "common.h"
#include "bf3_weights.h"

namespace afd {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(256) void bf3_weights(const float* __restrict__ w, __bf16* __restrict__ Wf, __bf16* __restrict__ Wd,
                                                   int Cin, int Cout) {
  bf3_weights_block(w, Wf, Wd, Cin, Cout, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// ---- geometry of a 128-pixel workgroup tile on an S x S map ------------------------------------------------------------
template <int S> struct BfGeo {
  static constexpr int TP = 128;
  static constexpr int IPT = S * S >= TP ? 1 : TP / (S * S);         // images per tile
  static constexpr int R = S * S >= TP ? TP / S : S;                 // output rows per image in the tile
  static constexpr int TPI = S * S >= TP ? S * S / TP : 1;           // tiles per image
  static constexpr int Wp = S + 2, IMG = (R + 2) * Wp, NPIX = IPT * IMG;
  static constexpr int NPP = (NPIX + 15) / 16 * 16;                  // channel-group stride = whole 64-bank rows: the four groups of a b128 read never collide
  static constexpr int KG = 4;                                       // 8-channel groups per 32-channel chunk
  static constexpr int TASKS = KG * NPIX, NE = (TASKS + 255) / 256;  // staging tasks (pixel record, channel group) per thread
};

template <int S, int NBLK>
__global__ __launch_bounds__(256, 2) void conv_bf3(const float* __restrict__ x, const bf8* __restrict__ Wp,
                                                   const float* __restrict__ bias, const float* __restrict__ res,
                                                   float* __restrict__ y, int B, int K, int N, int act) {
  using G = BfGeo<S>;
  // v_mfma_f32_16x16x32_bf16: lane (row / column = l & 15, k group = l >> 4) holds k = 8 (l >> 4) + j -- one 16-byte
  // record of the weight image (A) or of the LDS pixel image (B); accumulator: column = l & 15, row = 4 (l >> 4) + reg.
  // (The 16x16x32 shape holds a higher clock than 32x32x16 under this chip's power management: MI355X_MICROARCH.md.)
  constexpr int PSW = 2 * NBLK, NPIX = G::NPIX, NPP = G::NPP, NE = G::NE;   // 16-pixel sub-blocks per wave: 4 / NBLK waves share 128 pixels
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  bf8* Xp = reinterpret_cast<bf8*>(smem_raw);                        // [piece 3][kg 4][NPP] records
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kgl = lane >> 4, l15 = lane & 15;
  const int nb = wv % NBLK, pgrp = wv / NBLK;                        // this wave's 32-channel block / its group of pixel sub-blocks
  const int tile = blockIdx.x, n0 = (blockIdx.y * NBLK + nb) * 32;
  const int img0 = G::IPT > 1 ? tile * G::IPT : tile / G::TPI;
  const int row0 = G::IPT > 1 ? 0 : (tile % G::TPI) * G::R;
  const int HW = S * S;

  // ---- staging plan: task e of this thread = (record ridx, channel group kg)
  int s_src[NE];                 // element offset of channel (8 kg) at the record's pixel inside image img0 (or -1: zero)
  int s_dst[NE];                 // record index kg * NPP + ridx (or -1: no task)
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int t = tid + 256 * e;
    s_src[e] = -1; s_dst[e] = -1;
    if (t < G::TASKS) {
      const int kg = t / NPIX, ridx = t - kg * NPIX;
      const int i = ridx / G::IMG, rem = ridx - i * G::IMG, rr = rem / G::Wp, cc = rem - rr * G::Wp;
      const int yy = row0 + rr - 1, xx = cc - 1, b = img0 + i;
      s_dst[e] = kg * NPP + ridx;
      if (yy >= 0 && yy < S && xx >= 0 && xx < S && b < B) s_src[e] = ((i * K + 8 * kg) * S + yy) * S + xx;
    }
  }
  const float* xb = x + (long)img0 * K * HW;
  float xr[NE][8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const float* p = xb + (long)k0 * HW + (s_src[e] >= 0 ? s_src[e] : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[e][j] = s_src[e] >= 0 ? p[(long)j * HW] : 0.f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (s_dst[e] < 0) continue;
      bf8 p0, p1, p2;
#pragma unroll
      for (int j = 0; j < 8; ++j) { __bf16 a, b, c; bf3_split(xr[e][j], a, b, c); p0[j] = a; p1[j] = b; p2[j] = c; }
      Xp[s_dst[e]] = p0;
      Xp[G::KG * NPP + s_dst[e]] = p1;
      Xp[2 * G::KG * NPP + s_dst[e]] = p2;
    }
  };

  // ---- B fragment base per 16-pixel sub-block: record of (pixel, tap (0,0)) in this lane's channel group
  int bbase[PSW];
#pragma unroll
  for (int ps = 0; ps < PSW; ++ps) {
    const int q = (pgrp * PSW + ps) * 16 + l15;
    const int i = q / (G::R * S), rem = q - i * (G::R * S), r = rem / S, c = rem - r * S;
    bbase[ps] = kgl * NPP + i * G::IMG + r * G::Wp + c;
  }
  f32x4 acc[2][PSW];
#pragma unroll
  for (int ns = 0; ns < 2; ++ns)
#pragma unroll
    for (int ps = 0; ps < PSW; ++ps) acc[ns][ps] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- A fragments: records (tap * K/8 + k/8) * N + n, k/8 = 4 ks + (l >> 4), n = n0 + 16 ns + (l & 15)
  const int KS = K >> 5, nit = KS * 9;                               // (k-step of 32 channels, tap) groups
  const long pstride = (long)9 * (K >> 3) * N;                       // records per piece
  const bf8* wl = Wp + (long)kgl * N + n0 + l15;
  auto a_load = [&](bf8 (&a)[2][3], int it) {
    const int ks = it / 9, tap = it - ks * 9;
    const long r = ((long)tap * (K >> 3) + 4 * ks) * N;
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) { a[ns][0] = wl[r + 16 * ns]; a[ns][1] = wl[pstride + r + 16 * ns]; a[ns][2] = wl[2 * pstride + r + 16 * ns]; }
  };
#ifndef BF3_ABL
#define BF3_ABL 0          // ablation bits (tools/micro/bf3_abl.hip): 1 no MFMA, 2 no A loads after the first, 4 no x staging after the first chunk, 8 no B reads after the first
#endif
  // A queue, AQ groups deep: vector-memory results return in order, so a fragment requested after the next chunk's x loads
  // (HBM latency) cannot be used before those have landed
  constexpr int AQ = NBLK == 4 ? 1 : 2;                              // (register budget: 64 accumulators + 24 per queued group)
  bf8 aq[AQ][2][3];
#pragma unroll
  for (int d = 0; d < AQ; ++d)
    if (d < nit) a_load(aq[d], d);

  fetch(0);
  int it = 0;
  for (int k0 = 0; k0 < K; k0 += 32) {
    __syncthreads();                                                  // the previous chunk's fragment reads are done
    if (!(BF3_ABL & 4) || k0 == 0) commit();
    __syncthreads();
    // B fragments one (tap, pixel sub-block) unit ahead of the multiplies; the order is pinned (the scheduler would
    // otherwise hoist every read of the unrolled loop to the top and spill)
    bf8 bc[3], bn[3];
    bc[0] = Xp[bbase[0]]; bc[1] = Xp[G::KG * NPP + bbase[0]]; bc[2] = Xp[2 * G::KG * NPP + bbase[0]];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      bf8 a[2][3];
#pragma unroll
      for (int ns = 0; ns < 2; ++ns) { a[ns][0] = aq[0][ns][0]; a[ns][1] = aq[0][ns][1]; a[ns][2] = aq[0][ns][2]; }
#pragma unroll
      for (int d = 0; d + 1 < AQ; ++d)
#pragma unroll
        for (int ns = 0; ns < 2; ++ns) { aq[d][ns][0] = aq[d + 1][ns][0]; aq[d][ns][1] = aq[d + 1][ns][1]; aq[d][ns][2] = aq[d + 1][ns][2]; }
      if (it + AQ < nit && !(BF3_ABL & 2)) a_load(aq[AQ - 1], it + AQ);
      ++it;
      if (tap == 0 && k0 + 32 < K && !(BF3_ABL & 4)) fetch(k0 + 32);  // in flight during the multiplies, behind the next tap's A request
#pragma unroll
      for (int ps = 0; ps < PSW; ++ps) {
        const int u = tap * PSW + ps + 1;                            // the next unit
        if (u < 9 * PSW && !(BF3_ABL & 8)) {
          const int tn = u / PSW, pn = u % PSW;
          const int o = bbase[pn] + (tn / 3) * G::Wp + (tn % 3);
          bn[0] = Xp[o]; bn[1] = Xp[G::KG * NPP + o]; bn[2] = Xp[2 * G::KG * NPP + o];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (BF3_ABL & 1) {
          asm volatile("" :: "v"(a[0][0]), "v"(a[0][1]), "v"(a[0][2]), "v"(a[1][0]), "v"(bc[0]), "v"(bc[1]), "v"(bc[2]));
        } else {
          // (term-major order -- consecutive MFMAs into different accumulators -- measured no faster and costs the 4-block
          // form 12 spilled registers: back-to-back accumulation into one register block is free on this pipe)
#pragma unroll
          for (int ns = 0; ns < 2; ++ns) {
            acc[ns][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ns][0], bc[0], acc[ns][ps], 0, 0, 0);
            acc[ns][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ns][1], bc[0], acc[ns][ps], 0, 0, 0);
            acc[ns][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ns][0], bc[1], acc[ns][ps], 0, 0, 0);
            acc[ns][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ns][2], bc[0], acc[ns][ps], 0, 0, 0);
            acc[ns][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ns][1], bc[1], acc[ns][ps], 0, 0, 0);
            acc[ns][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ns][0], bc[2], acc[ns][ps], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(BF3_ABL & 8)) { bc[0] = bn[0]; bc[1] = bn[1]; bc[2] = bn[2]; }
      }
    }
  }

  // ---- epilogue: accumulator rows = output channels n0 + 16 ns + 4 (l >> 4) + reg, column = this lane's pixel
#pragma unroll
  for (int ps = 0; ps < PSW; ++ps) {
    const int q = (pgrp * PSW + ps) * 16 + l15;
    const int i = q / (G::R * S), rem = q - i * (G::R * S), r = rem / S, c = rem - r * S;
    const int b = img0 + i;
    if (b >= B) continue;
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) {
      const int nn = n0 + 16 * ns + 4 * kgl;
      const long o = ((long)b * N + nn) * HW + (row0 + r) * S + c;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float v = acc[ns][ps][rg];
        if (bias) v += bias[nn + rg];
        if (act == 1) v = gelu_erf(v);
        if (res) v += res[o + (long)rg * HW];
        y[o + (long)rg * HW] = v;
      }
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static int g_bf3_mode = 0;          // afd_debug_conv_path 80 / 81 / 82: by the rule / off / wherever the shape is covered
void bf3_set_mode(int m) { g_bf3_mode = m; }
static int g_direct_bf3 = 0;        // afd_debug_conv_path 76 / 77: the direct form is f16x2 (h2.hip, default) / bf16x3 (this file)
void direct_form_set(int m) { g_direct_bf3 = m; }
bool direct_form_is_bf3() { return g_direct_bf3 != 0; }
static int g_bf3_nblk = 0;          // tools/micro/bf3_abl.hip: force the channel blocks per workgroup (0 = by the rule)
void bf3_set_nblk(int n) { g_bf3_nblk = n; }
// channel blocks per workgroup: as many as divide N while the launch still has two workgroups per CU (four only on 16 x 16
// maps: the 8 x 8 and 32 x 32 staging plans hold one more task per thread and the 4-block form would spill)
static int bf3_nblk(long tiles, int N, int S) {
  if (g_bf3_nblk && N % (32 * g_bf3_nblk) == 0 && (g_bf3_nblk < 4 || S == 16)) return g_bf3_nblk;
  for (int nb = S == 16 ? 4 : 2; nb > 1; nb >>= 1)
    if (N % (32 * nb) == 0 && tiles * (N / (32 * nb)) >= 512) return nb;
  return 1;
}

// does a direct matrix-core kernel take the layer (reduction width K, N output channels)?  0 = no, 1 = the 128-pixel tile
// kernel (conv_h2 / conv_bf3), 2 = the split-K small-map kernel (conv_h2_sk: f16x2 arithmetic only; afd_debug_conv_path 74 / 75
// = by this rule (default) / never)
bool h2_sk_ok(int B, int K, int N, int H, int W, bool force);          // h2.hip
static int g_sk_mode = 0;           // afd_debug_conv_path 74 / 75 / 73: by the rule / never / wherever the shape is covered (ahead of the tile kernel)
void h2_sk_set_mode(int m) { g_sk_mode = m; }
int direct_plan(int B, int K, int N, int H, int W) {
  if (g_bf3_mode == 1) return 0;
  if (!g_direct_bf3 && g_sk_mode == 2 && h2_sk_ok(B, K, N, H, W, true)) return 2;
  const bool sk = !g_direct_bf3 && g_sk_mode != 1 && h2_sk_ok(B, K, N, H, W, g_bf3_mode == 2);
  // (4 x 4 maps on the tile kernel: 8 images per tile, 32 tiles x N / 32 = 128-256 workgroups of one wave per SIMD: measured
  // 20 / 41 us for 128->128 / 256->256 in round 2 -- not instantiated; the split-K kernel takes them)
  if (H != W || (W != 8 && W != 16 && W != 32)) return sk ? 2 : 0;
  if (K % 32 || N % 32 || K < 32) return 0;
  if ((long)B * K * H * W >= (1L << 31) || (long)B * N * H * W >= (1L << 31)) return 0;
  if (g_bf3_mode == 2) return 1;
  const long tiles = ((long)B * H * W + 127) / 128;
  const int nblk = bf3_nblk(tiles, N, W);
  if (tiles * (N / (32 * nblk)) >= 512) return 1;                     // two workgroups per CU (measured: below that the small-map forms win)
  return sk ? 2 : 0;
}
bool bf3_ok(int B, int K, int N, int H, int W) { return direct_plan(B, K, N, H, W) != 0; }
size_t bf3_weight_bytes(int Cin, int Cout) { return (size_t)54 * Cin * Cout; }

void bf3_weights_launch(const float* w, void* Wf, void* Wd, int Cin, int Cout, hipStream_t s) {
  hipLaunchKernelGGL(bf3_weights, dim3((unsigned)((Cin * Cout / 64 + 3) / 4)), dim3(256), 0, s, w, static_cast<__bf16*>(Wf),
                     static_cast<__bf16*>(Wd), Cin, Cout);
}

template <int S, int NBLK>
static void bf3_launch_t(const float* x, const void* Wp, const float* bias, const float* res, float* y, int B, int K, int N, int act,
                         hipStream_t s) {
  using G = BfGeo<S>;
  const size_t lds = (size_t)3 * G::KG * G::NPP * 16;
  const long tiles = G::IPT > 1 ? ((long)B + G::IPT - 1) / G::IPT : (long)B * G::TPI;
  hipLaunchKernelGGL((conv_bf3<S, NBLK>), dim3((unsigned)tiles, (unsigned)(N / (32 * NBLK))), dim3(256), lds, s, x,
                     static_cast<const bf8*>(Wp), bias, res, y, B, K, N, act);
}
// x (B,K,S,S), Wp = the bf16x3 weight image for (K -> N) -> y (B,N,S,S)
void bf3_conv(const float* x, const void* Wp, const float* bias, const float* res, float* y, int B, int K, int N, int S, int act,
              hipStream_t s) {
  const int nblk = bf3_nblk(((long)B * S * S + 127) / 128, N, S);
#define AFD_BF3(S_)                                                                      \
  if (nblk == 4 && S_ == 16) bf3_launch_t<16, 4>(x, Wp, bias, res, y, B, K, N, act, s);  \
  else if (nblk == 2) bf3_launch_t<S_, 2>(x, Wp, bias, res, y, B, K, N, act, s);         \
  else bf3_launch_t<S_, 1>(x, Wp, bias, res, y, B, K, N, act, s)
  if (S == 32) { AFD_BF3(32); } else if (S == 16) { AFD_BF3(16); } else { AFD_BF3(8); }
#undef AFD_BF3
}

}  // namespace afd

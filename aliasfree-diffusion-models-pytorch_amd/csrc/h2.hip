// h2.hip -- F5: 3x3 (pad 1) convolution forward / dgrad as a DIRECT implicit GEMM on the fp16 matrix cores at fp32-class
// accuracy with TWO-piece splits and online power-of-two scaling (round 3; h2_common.h has the arithmetic and its error bound).
//
// Same tiling as round 2's bf16x3 kernel (bf3.hip, kept for A/B: afd_debug_conv_path 76 / 77): a workgroup owns NBLK x 32
// output channels x 128 output pixels (whole rows of one image, or whole images); per chunk of 32 input channels the haloed
// input pixels go global -> registers -> scale -> split -> LDS as Xp[piece 2][k/8][pixel][8 fp16]; a B fragment is one
// ds_read_b128 at (pixel + tap offset); A fragments come straight from global memory / L2 out of the weight image
// Wp[piece 2][tap][k/8][n][8 fp16].  THREE v_mfma_f32_16x16x32_f16 per (tap, chunk, 16 x 16 block) instead of six.
//
// Online scale: before a staged chunk is split, the workgroup takes the maximum magnitude of the values its threads hold
// (a DPP wave maximum, four floats through LDS, folded into the barrier that already separates the previous chunk's fragment
// reads from this chunk's stores), and if the chunk would leave fp16's range under the current scale, lowers the scale and
// multiplies the accumulators by the ratio.  The epilogue multiplies by 1 / (s_x s_w[n]).  All factors are powers of two.
#include <algorithm>
#include <cstdint>
#include "common.h"
#include "h2_common.h"

namespace afd {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// Ablation builds of conv_h2 (tools/abl_conv.sh: one component of the main loop removed per variant library, timings only --
// the results are wrong): 1 no MFMAs, 2 no A loads in the loop, 4 no x fetch in the loop, 8 no split / LDS stores after the
// first chunk, 16 no B reads in the loop, 32 no epilogue stores.  0 (the product) compiles to exactly the code without them.
#ifndef AFD_H2_ABL
#define AFD_H2_ABL 0
#endif
#ifndef AFD_H2_AQ
#define AFD_H2_AQ 2
#endif
#ifndef AFD_H2_PRIO
#define AFD_H2_PRIO 0
#endif
#ifndef AFD_H2_SLEEP
#define AFD_H2_SLEEP 0
#endif
#ifndef AFD_H2_LDSPAD
#define AFD_H2_LDSPAD 0                // (experiment: extra dynamic LDS per workgroup of conv_h2l, to force one workgroup per CU)
#endif
#ifndef AFD_H2_STAMP
#define AFD_H2_STAMP 0                 // timing build of conv_h2l: `res` is a buffer of 16 s_memtime stamps per workgroup (tools/h2_stamps.py)
#endif

__global__ __launch_bounds__(256) void h2_wscale(const float* __restrict__ w, void* Wf, void* Wd, int Cin, int Cout) {
  h2_wscale_rows(w, Wf, Wd, Cin, Cout, blockIdx.x * 4 + (threadIdx.x >> 6), gridDim.x * 4, threadIdx.x & 63);
}
__global__ __launch_bounds__(256) void h2_weights(const float* __restrict__ w, void* Wf, void* Wd, int Cin, int Cout) {
  h2_weights_block(w, Wf, Wd, Cin, Cout, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// ---- geometry of a 128-pixel workgroup tile on an S x S map (as bf3.hip) -------------------------------------------------
template <int S> struct H2Geo {
  static constexpr int TP = 128;
  static constexpr int IPT = S * S >= TP ? 1 : TP / (S * S);         // images per tile
  static constexpr int R = S * S >= TP ? TP / S : S;                 // output rows per image in the tile
  static constexpr int TPI = S * S >= TP ? S * S / TP : 1;           // tiles per image
  static constexpr int Wp = S + 2, IMG = (R + 2) * Wp, NPIX = IPT * IMG;
  static constexpr int NPP = (NPIX + 15) / 16 * 16;                  // channel-group stride = whole 64-bank rows: the four groups of a b128 read never collide
  static constexpr int KG = 4;                                       // 8-channel groups per 32-channel chunk
  static constexpr int TASKS = KG * NPIX, NE = (TASKS + 255) / 256;  // staging tasks (pixel record, channel group) per thread
  // pixel of MFMA column l15 of 16-pixel sub-block `blk` of the tile -> (image i, row r, column c).  16 x 16 / 32 x 32 maps: 16
  // consecutive pixels of a row (16 consecutive LDS records: all 64 banks once).  8 x 8 maps: two rows of 8 -- rows r and r + 4,
  // not r and r + 1: with the haloed row stride of 10 records the second row then starts 40 records = 8 (mod 16) behind the
  // first, so the two runs of 8 records cover the 64 banks exactly once (rows r, r + 1 collide on 2 of 16 lanes: PMC showed
  // 2.5 conflict cycles per LDS instruction on these layers)
  __device__ static void pixel(int blk, int l15, int& i, int& r, int& c) {
    if (S == 8) { i = blk >> 2; r = (blk & 3) + 4 * (l15 >> 3); c = l15 & 7; }
    else { const int q = blk * 16 + l15; i = q / (R * S); const int rem = q - i * (R * S); r = rem / S; c = rem - r * S; }
  }
};

template <int S, int NBLK>
__global__ __launch_bounds__(256, 2) void conv_h2(const float* __restrict__ x, const h8* __restrict__ Wp,
                                                  const float* __restrict__ bias, const float* __restrict__ res,
                                                  float* __restrict__ y, int B, int K, int N, int act) {
  using G = H2Geo<S>;
  // v_mfma_f32_16x16x32_f16: lane (row / column = l & 15, k group = l >> 4) holds k = 8 (l >> 4) + j -- one 16-byte
  // record of the weight image (A) or of the LDS pixel image (B); accumulator: column = l & 15, row = 4 (l >> 4) + reg.
  constexpr int PSW = 2 * NBLK, NPIX = G::NPIX, NPP = G::NPP, NE = G::NE;   // 16-pixel sub-blocks per wave: 4 / NBLK waves share 128 pixels
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  h8* Xp = reinterpret_cast<h8*>(smem_raw);                          // [piece 2][kg 4][NPP] records
  float* wmax = reinterpret_cast<float*>(smem_raw + (size_t)2 * G::KG * NPP * 16);   // the four waves' chunk maxima
#if AFD_H2_SLEEP
  if (__builtin_amdgcn_s_getreg(6148) & 1) __builtin_amdgcn_s_sleep(AFD_H2_SLEEP);   // (experiment: a staggered start instead)
#endif
#if AFD_H2_PRIO
  // the two waves that share a SIMD belong to two workgroups that start together and walk the same phases in lockstep; giving
  // them different priorities (by wave slot, HW_ID[3:0]) lets one run its multiplies while the other stages
  if (__builtin_amdgcn_s_getreg(6148) & 1) __builtin_amdgcn_s_setprio(AFD_H2_PRIO);
#endif
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
  // the chunk's loads.  32 x 32 maps: every lane issues every load (halo / padding tasks read element 0 and are zeroed in
  // chunk_amax) -- the per-task exec masks cost a branch and a mask save / restore per pair of loads (27.6 -> 26.1, 47.3 -> 44.6 us
  // on the 32-channel layers).  Smaller maps keep the masked form: a third of their tasks is padding and skipping those loads
  // is worth more (8 x 8: 53.3 against 59.1 us at 256 -> 256).  Buffer loads (scalar descriptor + offset, one instruction per
  // load) were measured too: no faster for x, 5-15 % slower for the weight fragments
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const float* p = xb + (long)k0 * HW + (s_src[e] >= 0 ? s_src[e] : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[e][j] = (S == 32 || s_src[e] >= 0) ? p[(long)j * HW] : 0.f;
    }
  };
  auto chunk_amax = [&]() {                                           // this wave's share of the staged chunk's max |x| -> wmax[wv]
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int j = 0; j < 8; ++j) { xr[e][j] = s_src[e] >= 0 ? xr[e][j] : 0.f; m = fmaxf(m, fabsf(xr[e][j])); }
    m = wave_amax(m);
    if (lane == 0) wmax[wv] = m;
  };
  auto commit = [&](float s) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (s_dst[e] < 0) continue;
      h8 p0, p1;
      h2_split8(xr[e], s, p0, p1);
      Xp[s_dst[e]] = p0;
      Xp[G::KG * NPP + s_dst[e]] = p1;
    }
  };

  // ---- B fragment base per 16-pixel sub-block: record of (pixel, tap (0,0)) in this lane's channel group
  int bbase[PSW];
#pragma unroll
  for (int ps = 0; ps < PSW; ++ps) {
    int i, r, c;
    G::pixel(pgrp * PSW + ps, l15, i, r, c);
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
  const h8* wl = Wp + (long)kgl * N + n0 + l15;
  auto a_load = [&](h8 (&a)[2][2], int it) {
    const int ks = it / 9, tap = it - ks * 9;
    const long r = ((long)tap * (K >> 3) + 4 * ks) * N;
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) { a[ns][0] = wl[r + 16 * ns]; a[ns][1] = wl[pstride + r + 16 * ns]; }
  };
  // A queue, AQ groups deep: vector-memory results return in order, so a fragment requested after the next chunk's x loads
  // (HBM latency) cannot be used before those have landed
  constexpr int AQ = AFD_H2_AQ;
  h8 aq[AQ][2][2];
#pragma unroll
  for (int d = 0; d < AQ; ++d)
    if (d < nit) a_load(aq[d], d);

  float sx = __uint_as_float(kH2ScaleCapBits);                        // the running scale of x: the cap until a chunk says otherwise
  fetch(0);
  int it = 0;
  for (int k0 = 0; k0 < K; k0 += 32) {
    const bool stage = !(AFD_H2_ABL & 8) || k0 == 0;
    if (stage) chunk_amax();
    __syncthreads();                                                  // the previous chunk's fragment reads are done; the four maxima are visible
    if (stage) {
      const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      const float sn = fminf(sx, h2_scale_for(m));
      if (sn != sx) {                                                 // (uniform) this chunk would overflow: lower the scale, carry the sums over
        const float f = sn * h2_inv_pow2(sx);
#pragma unroll
        for (int ns = 0; ns < 2; ++ns)
#pragma unroll
          for (int ps = 0; ps < PSW; ++ps) acc[ns][ps] *= f;
        sx = sn;
      }
    }
    if (stage) commit(sx);
    __syncthreads();
    // B fragments one (tap, pixel sub-block) unit ahead of the multiplies; the order is pinned (the scheduler would
    // otherwise hoist every read of the unrolled loop to the top and spill)
    h8 bc[2], bn[2];
    bc[0] = Xp[bbase[0]]; bc[1] = Xp[G::KG * NPP + bbase[0]];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      h8 a[2][2];
#pragma unroll
      for (int ns = 0; ns < 2; ++ns) { a[ns][0] = aq[0][ns][0]; a[ns][1] = aq[0][ns][1]; }
#pragma unroll
      for (int d = 0; d + 1 < AQ; ++d)
#pragma unroll
        for (int ns = 0; ns < 2; ++ns) { aq[d][ns][0] = aq[d + 1][ns][0]; aq[d][ns][1] = aq[d + 1][ns][1]; }
      if (it + AQ < nit && !(AFD_H2_ABL & 2)) a_load(aq[AQ - 1], it + AQ);
      ++it;
      if (tap == 0 && k0 + 32 < K && !(AFD_H2_ABL & 4)) fetch(k0 + 32);                    // in flight during the multiplies, behind the next tap's A request
#pragma unroll
      for (int ps = 0; ps < PSW; ++ps) {
        const int u = tap * PSW + ps + 1;                            // the next unit
        if (u < 9 * PSW && !(AFD_H2_ABL & 16)) {
          const int tn = u / PSW, pn = u % PSW;
          const int o = bbase[pn] + (tn / 3) * G::Wp + (tn % 3);
          bn[0] = Xp[o]; bn[1] = Xp[G::KG * NPP + o];
        } else if (AFD_H2_ABL & 16) { bn[0] = bc[0]; bn[1] = bc[1]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ns = 0; ns < 2; ++ns) {
          if (AFD_H2_ABL & 1) { asm volatile("" :: "v"(a[ns][0]), "v"(a[ns][1]), "v"(bc[0]), "v"(bc[1])); continue; }
          acc[ns][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ns][0], bc[0], acc[ns][ps], 0, 0, 0);
          acc[ns][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ns][1], bc[0], acc[ns][ps], 0, 0, 0);
          acc[ns][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ns][0], bc[1], acc[ns][ps], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        bc[0] = bn[0]; bc[1] = bn[1];
      }
    }
  }

  // ---- epilogue: accumulator rows = output channels n0 + 16 ns + 4 (l >> 4) + reg, column = this lane's pixel.  Two 16-pixel
  // sub-blocks at a time through a wave-private LDS tile [32 channels][36], so that a lane leaves with four consecutive pixels
  // of one channel: 4 dwordx4 stores per pair instead of 16 dword stores (the ablation above: the dword store tail was 14 % of
  // the kernel; the same change took the LDS-fed kernel's tail from 7.0 to 3.5 thousand cycles)
  if ((AFD_H2_ABL & 32) && act != 77) return;
  const float* isw = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(Wp) + h2_scale_offset_bytes(K, N));
  const float isx = h2_inv_pow2(sx);
  __syncthreads();                                                    // every wave is done with the X image
  float* T = reinterpret_cast<float*>(smem_raw) + wv * (32 * 36);
  const int cn = lane >> 3, cm = lane & 7;                            // store pass: channel 8 it + cn, pixels 4 (cm & 3) .. + 3 of sub-block cm >> 2 of the pair
  float sw[4], bv[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) { sw[it] = isw[n0 + 8 * it + cn]; bv[it] = bias ? bias[n0 + 8 * it + cn] : 0.f; }
#pragma unroll
  for (int pp = 0; pp < PSW / 2; ++pp) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ns = 0; ns < 2; ++ns)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) T[(16 * ns + 4 * kgl + rg) * 36 + 16 * h + l15] = acc[ns][2 * pp + h][rg] * isx;
    int i, r, c;
    G::pixel(pgrp * PSW + 2 * pp + (cm >> 2), 4 * (cm & 3), i, r, c);
    const int b = img0 + i;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int nn = n0 + 8 * it + cn;
      f32x4 v = *reinterpret_cast<const f32x4*>(T + (8 * it + cn) * 36 + 4 * cm);
      if (b >= B) continue;
      const long o = ((long)b * N + nn) * HW + (row0 + r) * S + c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = v[e] * sw[it] + bv[it];
        if (act == 1) u = gelu_erf(u);
        v[e] = u;
      }
      if (res) v += *reinterpret_cast<const f32x4*>(res + o);
      *reinterpret_cast<f32x4*>(y + o) = v;
    }
  }
}

// ---- the LDS-fed tile kernel -------------------------------------------------------------------------------------------------
// Ablation of conv_h2 above (tools/abl_conv.sh): its components ADD -- the multiplies are 25 % of its
// time, the per-lane weight-fragment loads 16 % (a wave re-reads from L2, through the 64 B/clk vector-memory path, fragments
// that the other waves of the workgroup read too: at 32 channels x 32 pixels per wave the path is oversubscribed 2x), the
// output stores 14 % (64-byte runs).  This form feeds BOTH operands from LDS:
//   * the weight image of one filter ROW (3 taps x 32 input channels x NT output channels, both pieces: 12 KB per 32 output
//     channels) arrives by LDS-DMA (global_load_lds_dwordx4, no registers, one 1-KB piece per wave-instruction) into one of two
//     buffers, one row ahead of the multiplies; a fragment is one conflict-free ds_read_b128;
//   * v_mfma_f32_32x32x16_f16: the same operand bytes per flop, but the instruction holds the SIMD's vector issue for 8 of its
//     32 cycles instead of 8 of 16 (the staging arithmetic of the partner wave gets through), and an accumulator register holds
//     32 consecutive pixels of one channel: the output leaves in 128-byte runs.
// Per chunk of 32 input channels: [barrier, split + store X, barrier] row 0 [barrier] row 1 [barrier] row 2; every wait on the
// vector-memory counter is a vmcnt(0) at the end of a row (the DMA pieces of the next row and, in row 0, the next chunk's x).
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int S, int TP> struct H2LGeo {
  static constexpr int IPT = S * S >= TP ? 1 : TP / (S * S);         // images per tile
  static constexpr int R = S * S >= TP ? TP / S : S;                 // output rows per image in the tile
  static constexpr int TPI = S * S >= TP ? S * S / TP : 1;           // tiles per image
  static constexpr int Wp = S + 2, IMG = (R + 2) * Wp, NPIX = IPT * IMG;
  static constexpr int NPP = (NPIX + 15) / 16 * 16;
  static constexpr int KG = 4, TASKS = KG * NPIX, NE = (TASKS + 255) / 256;
  // pixel of MFMA column l31 of 32-pixel block `blk`.  8 x 8 maps: a block is rows {2h, 2h+4, 2h+1, 2h+5} of half h of an image,
  // so that every 16 lanes read rows r and r + 4 (40 records apart: the two runs of 8 records cover the 64 banks once)
  __device__ static void pixel(int blk, int l31, int& i, int& r, int& c) {
    if (S == 8) { i = blk >> 1; r = 2 * (blk & 1) + ((l31 >> 4) & 1) + 4 * ((l31 >> 3) & 1); c = l31 & 7; }
    else { const int q = blk * 32 + l31; i = q / (R * S); const int rem = q - i * (R * S); r = rem / S; c = rem - r * S; }
  }
};

template <int S, int TP, int NB>
__global__ __launch_bounds__(256, 2) void conv_h2l(const float* __restrict__ x, const h8* __restrict__ Wp,
                                                   const float* __restrict__ bias, const float* __restrict__ res,
                                                   float* __restrict__ y, int B, int K, int N, int act) {
  using G = H2LGeo<S, TP>;
  constexpr int NT = 32 * NB, PGR = 4 / NB, PB = TP / 32 / PGR;      // output channels per workgroup, pixel groups, 32-pixel blocks per wave
  constexpr int NPIX = G::NPIX, NPP = G::NPP, NE = G::NE;
  constexpr int XREC = 2 * G::KG * NPP;                              // records of the X image [piece 2][kg 4][NPP]
  constexpr int AROW = 2 * 3 * 4 * NT;                               // records of one weight row buffer [piece 2][tx 3][kg 4][NT]
  constexpr int CPW = AROW / 64 / 4;                                 // DMA pieces (64 records) per wave per row
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  h8* Xp = reinterpret_cast<h8*>(smem_raw);
  h8* Ab = Xp + XREC;                                                // two row buffers
  float* wmax = reinterpret_cast<float*>(Ab + 2 * AROW);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int nb = wv % NB, pgrp = wv / NB;
  const int tile = blockIdx.x, n0w = blockIdx.y * NT, n0 = n0w + 32 * nb;
  const int img0 = G::IPT > 1 ? tile * G::IPT : tile / G::TPI;
  const int row0 = G::IPT > 1 ? 0 : (tile % G::TPI) * G::R;
  const int HW = S * S;
#if AFD_H2_SLEEP
  if (((blockIdx.y * gridDim.x + blockIdx.x) >> 8) & 1) { for (int z = 0; z < AFD_H2_SLEEP / 100; ++z) __builtin_amdgcn_s_sleep(100); }   // (experiment: a staggered start of every second sweep of 256 workgroups)
#endif
#if AFD_H2_STAMP
  long* stamp = tid == 0 ? reinterpret_cast<long*>(const_cast<float*>(res)) + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 16 : nullptr;
  int si = 0;
  res = nullptr;
#define AFD_STAMP() do { if (stamp) stamp[si] = (long)__builtin_amdgcn_s_memtime(); ++si; } while (0)
  if (stamp) stamp[15] = __builtin_amdgcn_s_getreg(6148 | (15 << 11));   // HW_ID[15:0]: wave, simd, pipe, cu, sh, se
  AFD_STAMP();                                                        // 0: start
#else
#define AFD_STAMP() do { } while (0)
#endif

  // ---- staging plan (as conv_h2): task e of this thread = (record ridx, channel group kg)
  int s_src[NE], s_dst[NE];
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
  // EVERY lane issues EVERY load (halo / padding tasks read element 0 of the chunk and are zeroed at the chunk boundary): the
  // counted wait at the end of filter row 1 relies on exactly NE * 8 loads being younger than that row's DMA pieces
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const float* p = xb + (long)k0 * HW + (s_src[e] >= 0 ? s_src[e] : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[e][j] = p[(long)j * HW];
    }
  };
  auto chunk_amax = [&]() {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int j = 0; j < 8; ++j) { xr[e][j] = s_src[e] >= 0 ? xr[e][j] : 0.f; m = fmaxf(m, fabsf(xr[e][j])); }
    m = wave_amax(m);
    if (lane == 0) wmax[wv] = m;
  };
  auto commit = [&](float s) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (s_dst[e] < 0) continue;
      h8 p0, p1;
      h2_split8(xr[e], s, p0, p1);
      Xp[s_dst[e]] = p0;
      Xp[G::KG * NPP + s_dst[e]] = p1;
    }
  };

  // ---- the weight rows by LDS-DMA: piece c of a row = 64 consecutive records of the buffer [piece][tx][kg][NT]; lane's source
  // record at (chunk 0, filter row 0).  NT = 64: c = (piece * 3 + tx) * 4 + kg, n = lane; NT = 32: c = (piece * 3 + tx) * 2 + kg / 2,
  // kg = 2 (c & 1) + (lane >> 5), n = lane & 31
  const int K8 = K >> 3;
  int soff[CPW];
#pragma unroll
  for (int d = 0; d < CPW; ++d) {
    const int c = wv * CPW + d;
    const int pt = NT == 64 ? c >> 2 : c >> 1, kg = NT == 64 ? c & 3 : 2 * (c & 1) + lh, n = NT == 64 ? lane : l31;
    const int pc = pt / 3, tx = pt - 3 * pc;
    soff[d] = ((pc * 9 + tx) * K8 + kg) * N + n0w + n;
  }
  auto dma_row = [&](int q) {                                         // q = 3 * chunk + filter row -> buffer q & 1
    const int ks = q / 3, ty = q - 3 * ks;
    const h8* src = Wp + (long)(ty * 3 * K8 + 4 * ks) * N;
    h8* dst = Ab + (q & 1) * AROW + wv * CPW * 64;
#pragma unroll
    for (int d = 0; d < CPW; ++d)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + soff[d]),
                                       (__attribute__((address_space(3))) void*)(dst + d * 64), 16, 0, 0);
  };
  // end of a filter row: this wave's DMA pieces have landed (all but the `younger` vector-memory operations issued after them),
  // then the workgroup's barrier -- a raw s_barrier: __syncthreads() would drain the x loads in flight with the pieces
#define AFD_ROW_END(younger)                                                                                              \
  do {                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    __builtin_amdgcn_s_waitcnt(((younger) & 15) | 0x0F70 | (((younger) >> 4) << 14));   /* vmcnt(younger) only */             \
    asm volatile("" ::: "memory");                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                                         \
    asm volatile("" ::: "memory");                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  } while (0)

  dma_row(0);
  fetch(0);
  AFD_STAMP();                                                        // 1: first loads issued

  // ---- fragment bases
  int bbase[PB];
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    int i, r, c;
    G::pixel(pgrp * PB + pb, l31, i, r, c);
    bbase[pb] = lh * NPP + i * G::IMG + r * G::Wp + c;
  }
  const int abase = lh * NT + 32 * nb + l31;                          // + ((piece * 3 + tx) * 4 + 2 kh) * NT
  f32x16 acc[PB];
#pragma unroll
  for (int pb = 0; pb < PB; ++pb)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[pb][j] = 0.f;

  const int KS = K >> 5, nrows = 3 * KS;
  float sx = __uint_as_float(kH2ScaleCapBits);
  AFD_STAMP();                                                        // 2: plan done
  int q = 0;
  for (int k0 = 0; k0 < K; k0 += 32) {
    chunk_amax();                                                     // (waits for the chunk's x: vmcnt(0), so row 3 ks of the weights has landed too)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_sched_barrier(0);
    if (k0 < 64) AFD_STAMP();                                         // 3 / 8: the chunk's x and weight row have landed
    __syncthreads();
    {
      const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      const float sn = fminf(sx, h2_scale_for(m));
      if (sn != sx) {
        const float f = sn * h2_inv_pow2(sx);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) acc[pb] *= f;
        sx = sn;
      }
    }
    commit(sx);
    __syncthreads();
    if (k0 < 64) AFD_STAMP();                                         // 4 / 9: X committed
#pragma unroll
    for (int ty = 0; ty < 3; ++ty, ++q) {
      if (q + 1 < nrows) dma_row(q + 1);                              // into the buffer the row before this one was read from
      const bool prefetch = ty == 1 && k0 + 32 < K;                   // the next chunk's x: two rows of multiplies to land in
      if (prefetch) fetch(k0 + 32);
      const h8* Ar = Ab + (q & 1) * AROW + abase;
      // units (tx, kh, pb) of three multiplies; the fragments of the next unit are requested before the current one's multiplies
      // (the order is pinned: the scheduler would otherwise put every read right in front of its use)
      constexpr int U = 3 * 2 * PB;
      h8 ac[2], an[2], bc[2], bn[2];
      ac[0] = Ar[0]; ac[1] = Ar[3 * 4 * NT];
      bc[0] = Xp[bbase[0] + ty * G::Wp]; bc[1] = Xp[G::KG * NPP + bbase[0] + ty * G::Wp];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pb = u % PB;
        if (u + 1 < U) {
          const int un = u + 1, txn = un / (2 * PB), khn = (un / PB) % 2, pbn = un % PB;
          if (pbn == 0) { an[0] = Ar[(txn * 4 + 2 * khn) * NT]; an[1] = Ar[((3 + txn) * 4 + 2 * khn) * NT]; }
          const int o = bbase[pbn] + ty * G::Wp + txn + 2 * khn * NPP;
          bn[0] = Xp[o]; bn[1] = Xp[G::KG * NPP + o];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac[0], bc[0], acc[pb], 0, 0, 0);
        acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac[1], bc[0], acc[pb], 0, 0, 0);
        acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac[0], bc[1], acc[pb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < U) {
          if ((u + 1) % PB == 0) { ac[0] = an[0]; ac[1] = an[1]; }
          bc[0] = bn[0]; bc[1] = bn[1];
        }
      }
      if (ty == 0) AFD_ROW_END(0);                                    // (after row 2 the chunk boundary above does this)
      if (ty == 1) { if (prefetch) AFD_ROW_END(NE * 8); else AFD_ROW_END(0); }
      if (k0 < 64) AFD_STAMP();                                       // 5, 6, 7 / 10, 11, 12: rows done
    }
  }

  // ---- epilogue: accumulator register j = output channel n0 + 8 (j >> 2) + 4 (l >> 5) + (j & 3), column = this lane's pixel.
  // Through a wave-private LDS tile [32 channels][36] per pixel block, so that a lane leaves with four consecutive pixels of one
  // channel: 4 dwordx4 stores per block in 128-byte runs instead of 16 dword stores (the store tail of a workgroup was issue-bound:
  // 7 of its 33 thousand cycles on the 64 -> 64 @ 32 x 32 layer)
  const float* isw = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(Wp) + h2_scale_offset_bytes(K, N));
  const float isx = h2_inv_pow2(sx);
#if AFD_H2_STAMP
  si = 13;
#endif
  AFD_STAMP();                                                        // 13: loop done
  __syncthreads();                                                    // every wave is done with the X and weight images
  float* T = reinterpret_cast<float*>(smem_raw) + wv * (32 * 36);
  const int cn = lane >> 3, cm = lane & 7;                            // store pass: channel 8 it + cn, pixels 4 cm .. 4 cm + 3 of the block
  float sw[4], bv[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) { sw[it] = isw[n0 + 8 * it + cn]; bv[it] = bias ? bias[n0 + 8 * it + cn] : 0.f; }
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
    for (int j = 0; j < 16; ++j) T[(8 * (j >> 2) + 4 * lh + (j & 3)) * 36 + l31] = acc[pb][j] * isx;
    int i, r, c;
    G::pixel(pgrp * PB + pb, 4 * cm, i, r, c);
    const int b = img0 + i;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int nn = n0 + 8 * it + cn;
      f32x4 v = *reinterpret_cast<const f32x4*>(T + (8 * it + cn) * 36 + 4 * cm);
      if (b >= B) continue;
      const long o = ((long)b * N + nn) * HW + (row0 + r) * S + c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = v[e] * sw[it] + bv[it];
        if (act == 1) u = gelu_erf(u);
        v[e] = u;
      }
      if (res) v += *reinterpret_cast<const f32x4*>(res + o);
      *reinterpret_cast<f32x4*>(y + o) = v;
    }
  }
  AFD_STAMP();                                                        // 14: stores issued
#undef AFD_STAMP
}

// ---- the small maps (4 x 4, thin 8 x 8 layers): in-workgroup split-K ------------------------------------------------------
// At B = 256 a 4 x 4 map has 4096 pixels: 32 tiles of 128, too few workgroups for the kernel above (one wave per SIMD, every
// wave walking all of K: 20-40 us for a few GFLOP).  Here a workgroup owns 32 output channels x PSP slices of 32 pixels (two
// 4 x 4 images, or four rows of an 8 x 8 image: every halo cell outside the slice's rows is another slice's pixel or zero
// padding) and its four waves are KSP K-slices x PSP pixel slices that never meet in the loop: wave-private LDS image of its
// chunk, its own running scale, no barrier; the KSP partial sums are unscaled and added in wave order through LDS once, at
// the end.  Replaces round 1's fp32 Winograd split-K kernel (conv_wino_sk) wherever the f16x2 arithmetic is on.
template <int S> struct SkGeo {
  static constexpr int IPS = S == 4 ? 2 : 1;                          // images per 32-pixel slice
  static constexpr int R = 32 / (IPS * S);                            // rows per image in the slice (4 x 4: 4, 8 x 8: 4)
  static constexpr int Wp = S + 2, IMG = (R + 2) * Wp, NPIX = IPS * IMG, NPP = (NPIX + 15) / 16 * 16;
  static constexpr int KG = 4, TASKS = KG * NPIX, NE = (TASKS + 63) / 64;
};

template <int S, int KSP>
__global__ __launch_bounds__(256, 2) void conv_h2_sk(const float* __restrict__ x, const h8* __restrict__ Wp,
                                                     const float* __restrict__ bias, const float* __restrict__ res,
                                                     float* __restrict__ y, int B, int K, int N, int act, int nslices) {
  using G = SkGeo<S>;
  constexpr int PSP = 4 / KSP, NPP = G::NPP, NE = G::NE, HW = S * S;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kgl = lane >> 4, l15 = lane & 15;
  const int kw = wv % KSP, pw = wv / KSP;
  h8* Xw = reinterpret_cast<h8*>(smem_raw) + (size_t)wv * 2 * G::KG * NPP;     // this wave's [piece 2][kg 4][NPP] records
  float* red = reinterpret_cast<float*>(smem_raw + (size_t)4 * 2 * G::KG * NPP * 16);   // [wave][16][64] partial sums
  const int sl = blockIdx.x * PSP + pw, n0 = blockIdx.y * 32;
  const bool live = sl < nslices;
  const int img0 = S == 4 ? sl * 2 : sl / 2, row0 = S == 4 ? 0 : (sl & 1) * 4;

  int s_src[NE], s_dst[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int t = lane + 64 * e;
    s_src[e] = -1; s_dst[e] = -1;
    if (t < G::TASKS) {
      const int kg = t / G::NPIX, ridx = t - kg * G::NPIX;
      const int i = ridx / G::IMG, rem = ridx - i * G::IMG, rr = rem / G::Wp, cc = rem - rr * G::Wp;
      const int yy = row0 + rr - 1, xx = cc - 1, b = img0 + i;
      s_dst[e] = kg * NPP + ridx;
      if (live && yy >= 0 && yy < S && xx >= 0 && xx < S && b < B) s_src[e] = ((i * K + 8 * kg) * S + yy) * S + xx;
    }
  }
  const float* xb = x + (long)(live ? img0 : 0) * K * HW;
  float xr[NE][8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const float* p = xb + (long)k0 * HW + (s_src[e] >= 0 ? s_src[e] : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[e][j] = s_src[e] >= 0 ? p[(long)j * HW] : 0.f;
    }
  };
  int bbase[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int q = ps * 16 + l15;
    const int i = q / (G::R * S), rem = q - i * (G::R * S), r = rem / S, c = rem - r * S;
    bbase[ps] = kgl * NPP + i * G::IMG + r * G::Wp + c;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int ns = 0; ns < 2; ++ns)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) acc[ns][ps] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KS = K >> 5;
  const long pstride = (long)9 * (K >> 3) * N;
  const h8* wl = Wp + (long)kgl * N + n0 + l15;
  float sx = __uint_as_float(kH2ScaleCapBits);
  if (kw < KS) fetch(32 * kw);
  for (int ks = kw; ks < KS; ks += KSP) {
    // A fragments, three taps (one filter row) at a time, one row ahead of the multiplies; the first row is requested
    // before the LDS round trip of the staging
    h8 ac[3][2][2], an[3][2][2];
    auto a_row = [&](h8 (&a)[3][2][2], int ty) {
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const long r = ((long)(ty * 3 + tx) * (K >> 3) + 4 * ks) * N;
#pragma unroll
        for (int ns = 0; ns < 2; ++ns) { a[tx][ns][0] = wl[r + 16 * ns]; a[tx][ns][1] = wl[pstride + r + 16 * ns]; }
      }
    };
    a_row(ac, 0);
    {                                                                 // the chunk's magnitude -> this wave's scale (no one else reads it)
      float m = 0.f;
#pragma unroll
      for (int e = 0; e < NE; ++e)
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(xr[e][j]));
      const float sn = fminf(sx, h2_scale_for(wave_amax(m)));
      if (sn != sx) {
        const float f = sn * h2_inv_pow2(sx);
#pragma unroll
        for (int ns = 0; ns < 2; ++ns)
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) acc[ns][ps] *= f;
        sx = sn;
      }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (s_dst[e] < 0) continue;
      h8 p0, p1;
      h2_split8(xr[e], sx, p0, p1);
      Xw[s_dst[e]] = p0;
      Xw[G::KG * NPP + s_dst[e]] = p1;
    }
    if (ks + KSP < KS) fetch(32 * (ks + KSP));                        // the wave's next chunk in flight during the multiplies
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      if (ty < 2) a_row(an, ty + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int o = bbase[ps] + ty * G::Wp + tx;
          const h8 b0 = Xw[o], b1 = Xw[G::KG * NPP + o];
#pragma unroll
          for (int ns = 0; ns < 2; ++ns) {
            acc[ns][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ac[tx][ns][0], b0, acc[ns][ps], 0, 0, 0);
            acc[ns][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ac[tx][ns][1], b0, acc[ns][ps], 0, 0, 0);
            acc[ns][ps] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ac[tx][ns][0], b1, acc[ns][ps], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ty < 2) {
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
          for (int ns = 0; ns < 2; ++ns) { ac[tx][ns][0] = an[tx][ns][0]; ac[tx][ns][1] = an[tx][ns][1]; }
      }
    }
  }
  // ---- the K-slices' partial sums: unscaled (each wave had its own scale), added in wave order
  {
    const float isx = h2_inv_pow2(sx);
#pragma unroll
    for (int ns = 0; ns < 2; ++ns)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) red[(wv * 16 + (ns * 2 + ps) * 4 + rg) * 64 + lane] = acc[ns][ps][rg] * isx;
  }
  __syncthreads();
  if (kw != 0 || !live) return;
  const float* isw = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(Wp) + h2_scale_offset_bytes(K, N));
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int q = ps * 16 + l15;
    const int i = q / (G::R * S), rem = q - i * (G::R * S), r = rem / S, c = rem - r * S;
    const int b = img0 + i;
    if (b >= B) continue;
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) {
      const int nn = n0 + 16 * ns + 4 * kgl;
      const long o = ((long)b * N + nn) * HW + (row0 + r) * S + c;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < KSP; ++g) v += red[((pw * KSP + g) * 16 + (ns * 2 + ps) * 4 + rg) * 64 + lane];
        v *= isw[nn + rg];
        if (bias) v += bias[nn + rg];
        if (act == 1) v = gelu_erf(v);
        if (res) v += res[o + (long)rg * HW];
        y[o + (long)rg * HW] = v;
      }
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
// channel blocks per workgroup: as bf3.hip (as many as divide N while the launch keeps two workgroups per CU)
static int h2_nblk(long tiles, int N, int S) {
  static const long minwg = [] { const char* e = getenv("AFD_H2_MINWG"); return e ? atol(e) : 512L; }();   // tuning hook (two workgroups per CU)
  for (int nb = S == 16 ? 4 : 2; nb > 1; nb >>= 1)
    if (N % (32 * nb) == 0 && tiles * (N / (32 * nb)) >= minwg) return nb;
  return 1;
}

void h2_weights_launch(const float* w, void* Wf, void* Wd, int Cin, int Cout, hipStream_t s) {
  const int rows = (Wf ? Cout : 0) + (Wd ? Cin : 0);
  hipLaunchKernelGGL(h2_wscale, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, w, Wf, Wd, Cin, Cout);
  hipLaunchKernelGGL(h2_weights, dim3((unsigned)((Cin * Cout / 64 + 3) / 4)), dim3(256), 0, s, w, Wf, Wd, Cin, Cout);
}

template <int S, int NBLK>
static void h2_launch_t(const float* x, const void* Wp, const float* bias, const float* res, float* y, int B, int K, int N, int act,
                        hipStream_t s) {
  using G = H2Geo<S>;
  const size_t lds = (size_t)2 * G::KG * G::NPP * 16 + 16;
  const long tiles = G::IPT > 1 ? ((long)B + G::IPT - 1) / G::IPT : (long)B * G::TPI;
  hipLaunchKernelGGL((conv_h2<S, NBLK>), dim3((unsigned)tiles, (unsigned)(N / (32 * NBLK))), dim3(256), lds, s, x,
                     static_cast<const h8*>(Wp), bias, res, y, B, K, N, act);
}
template <int S, int KSP>
static void h2_sk_launch_t(const float* x, const void* Wp, const float* bias, const float* res, float* y, int B, int K, int N, int act,
                           hipStream_t s) {
  using G = SkGeo<S>;
  const size_t lds = (size_t)4 * 2 * G::KG * G::NPP * 16 + sizeof(float) * 4 * 16 * 64;
  const int nslices = (int)(((long)B * S * S + 31) / 32), psp = 4 / KSP;
  hipLaunchKernelGGL((conv_h2_sk<S, KSP>), dim3((unsigned)((nslices + psp - 1) / psp), (unsigned)(N / 32)), dim3(256), lds, s, x,
                     static_cast<const h8*>(Wp), bias, res, y, B, K, N, act, nslices);
}
// does the split-K small-map kernel take the layer?  4 x 4 and 8 x 8 maps, 32-channel blocks, at least two chunks of K, and a
// launch that the tile kernel above cannot fill (its rule: >= 512 workgroups) but this one can (>= 128)
bool h2_sk_ok(int B, int K, int N, int H, int W, bool force) {
  if (H != W || (W != 4 && W != 8) || K % 32 || N % 32 || K < 64) return false;
  if ((long)B * K * H * W >= (1L << 31) || (long)B * N * H * W >= (1L << 31)) return false;
  if (force) return true;
  const long slices = ((long)B * H * W + 31) / 32;
  const int ksp = K >= 128 ? 4 : 2;
  return ((slices + 4 / ksp - 1) / (4 / ksp)) * (N / 32) >= 128;
}
// ---- the LDS-fed tile kernel's launch: (tile pixels, 32-channel blocks per workgroup) = (128, 2) where N % 64 == 0 and the launch
// keeps two workgroups per CU, (256, 1) for the 32-output-channel layers of the big maps, (128, 1) otherwise
static int g_h2_lds = 1;                                              // afd_debug_conv_path 60 (by rule, default) / 61 (the register-fed kernel above) / 62, 63, 59 (tests: LDS-fed wherever covered, (128, 2) / (256, 1) / (128, 1) first)
void h2_lds_set(int m) { g_h2_lds = m; }
template <int S, int TP, int NB>
static int h2l_launch_t(const float* x, const void* Wp, const float* bias, const float* res, float* y, int B, int K, int N, int act,
                        hipStream_t s) {
  using G = H2LGeo<S, TP>;
  const size_t lds = (size_t)(2 * G::KG * G::NPP + 2 * 2 * 3 * 4 * 32 * NB) * 16 + 16 + AFD_H2_LDSPAD;
  if (int rc = lds_opt_in(conv_h2l<S, TP, NB>, lds)) return rc;
  const long tiles = G::IPT > 1 ? ((long)B + G::IPT - 1) / G::IPT : (long)B * G::TPI;
  hipLaunchKernelGGL((conv_h2l<S, TP, NB>), dim3((unsigned)tiles, (unsigned)(N / (32 * NB))), dim3(256), lds, s, x,
                     static_cast<const h8*>(Wp), bias, res, y, B, K, N, act);
  return AFD_OK;
}
template <int S>
static int h2l_launch(const float* x, const void* Wp, const float* bias, const float* res, float* y, int B, int K, int N, int act,
                      hipStream_t s) {
  const long px = (long)B * S * S;
  if (g_h2_lds != 4 && N % 64 == 0 && (g_h2_lds == 2 || (g_h2_lds == 1 && ((px + 127) / 128) * (N / 64) >= 512)))
    return h2l_launch_t<S, 128, 2>(x, Wp, bias, res, y, B, K, N, act, s);
  if constexpr (S >= 16) {
    if (g_h2_lds == 3 || (g_h2_lds == 1 && ((px + 255) / 256) * (N / 32) >= 512)) return h2l_launch_t<S, 256, 1>(x, Wp, bias, res, y, B, K, N, act, s);
  }
  return h2l_launch_t<S, 128, 1>(x, Wp, bias, res, y, B, K, N, act, s);
}

// x (B,K,S,S), Wp = the f16x2 weight image for (K -> N) -> y (B,N,S,S)
void h2_conv(const float* x, const void* Wp, const float* bias, const float* res, float* y, int B, int K, int N, int S, int act,
             hipStream_t s, bool small) {
  if (small) {
    if (S == 4) { if (K >= 128) h2_sk_launch_t<4, 4>(x, Wp, bias, res, y, B, K, N, act, s); else h2_sk_launch_t<4, 2>(x, Wp, bias, res, y, B, K, N, act, s); }
    else { if (K >= 128) h2_sk_launch_t<8, 4>(x, Wp, bias, res, y, B, K, N, act, s); else h2_sk_launch_t<8, 2>(x, Wp, bias, res, y, B, K, N, act, s); }
    return;
  }
  // by rule (measured per layer shape at B = 256, tools/abl_conv_bench.py 61 62): the LDS-fed kernel wins where a workgroup has ONE block
  // of 32 output channels on the 32 x 32 maps (every wave of the register-fed kernel then re-reads the same weight fragments:
  // 27.6 -> 25.3 and 47.3 -> 41.7 us), the register-fed one elsewhere (64 -> 64 @ 32 x 32: 63.5 against 67.9; 128 -> 128 @ 16 x 16: 48 against 63)
  if (g_h2_lds > 1 || (g_h2_lds == 1 && S == 32 && N % 64 != 0)) {
    if (S == 32) h2l_launch<32>(x, Wp, bias, res, y, B, K, N, act, s);
    else if (S == 16) h2l_launch<16>(x, Wp, bias, res, y, B, K, N, act, s);
    else h2l_launch<8>(x, Wp, bias, res, y, B, K, N, act, s);
    return;
  }
  const int nblk = h2_nblk(((long)B * S * S + 127) / 128, N, S);
#define AFD_H2(S_)                                                                      \
  if (nblk == 4 && S_ == 16) h2_launch_t<16, 4>(x, Wp, bias, res, y, B, K, N, act, s);  \
  else if (nblk == 2) h2_launch_t<S_, 2>(x, Wp, bias, res, y, B, K, N, act, s);         \
  else h2_launch_t<S_, 1>(x, Wp, bias, res, y, B, K, N, act, s)
  if (S == 32) { AFD_H2(32); } else if (S == 16) { AFD_H2(16); } else { AFD_H2(8); }
#undef AFD_H2
}

}  // namespace afd

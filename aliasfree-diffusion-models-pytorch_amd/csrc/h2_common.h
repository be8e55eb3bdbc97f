// h2_common.h -- the two-piece fp16 split with power-of-two block scaling ("f16x2") behind the direct 3x3 convolution
// kernels of round 3 (h2.hip, h2_wgrad.hip).
//
// Round 2 put the 3x3 products on the bf16 matrix pipe at fp32 accuracy with THREE bf16 pieces per value (8 mantissa bits
// each) and the six leading cross terms: 6 MFMAs per (tap, 32 channels, 16 x 16 block).  fp16 carries 11 mantissa bits, so TWO
// pieces hold 22 of an fp32 value's 24 bits and THREE cross terms (a1 b1, a2 b1, a1 b2) carry a product to 3 * 2^-22 worst
// case (2.4e-7 rms: the same class as the fp32 rounding of the operands themselves, 6e-8, and far inside the fp32
// accumulation error of a 288..2304-term dot product) -- half the matrix-pipe work of bf16x3, 3 instead of 5.5 vector
// instructions per split element, two thirds of the LDS traffic.
//
// What fp16 lacks is exponent range (normal numbers 6.1e-5 .. 65504), so every operand block is multiplied by a power of
// two s (exact) that brings its largest magnitude into [2^14, 2^15) before the split; the accumulators then hold
// s_a s_b * (the true sums) and the epilogue multiplies the two scales out again (exact).  Elements within 2^-17 of their
// block's maximum keep all 22 bits; smaller ones are represented with an ABSOLUTE error below 2^-39 of that maximum (fp16
// subnormals): invisible in any norm of the result.
//   weights      one scale per output row n of the weight image (max over its K x 9 taps), computed when the image is built
//                and stored behind it as 1 / s_n;
//   activations  ONLINE, like the running maximum of a streaming softmax: the workgroup (or wave) keeps one scale, looks at
//                the maximum of every chunk it stages BEFORE splitting it, and when a chunk would overflow lowers the scale
//                and multiplies its accumulators by the ratio (a power of two: exact) -- no pre-pass over the operand, no
//                per-tensor statistics from the producers.
#pragma once
#include <cstdint>
#include "common.h"

namespace afd {

using h8 = __attribute__((ext_vector_type(8))) _Float16;

// power of two s with m * s in [2^14, 2^15)  (m >= 0; m = 0 or below 2^-112: the cap 2^126; inf / nan: 2^-114, the value propagates)
constexpr uint32_t kH2ScaleCapBits = 253u << 23;       // 2^126: the running scale before any data has been seen
__device__ __forceinline__ float h2_scale_for(float m) {
  const int eb = (int)((__float_as_uint(m) >> 23) & 0xffu);
  int sb = 268 - eb;                                   // 127 + 14 - (eb - 127)
  sb = sb > 253 ? 253 : (sb < 1 ? 1 : sb);
  return __uint_as_float((uint32_t)sb << 23);
}
// 1 / s for a power of two s in [2^-126, 2^127]
__device__ __forceinline__ float h2_inv_pow2(float s) {
  return __uint_as_float((254u - (__float_as_uint(s) >> 23)) << 23);
}

// x * s = h1 + h2 + O(2^-22 |x s|): two fp16 pieces of 8 values (one 16-byte MFMA fragment each)
__device__ __forceinline__ void h2_split8(const float (&x)[8], float s, h8& p0, h8& p1) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = x[j] * s;
    const _Float16 a = (_Float16)v;
    p0[j] = a;
    p1[j] = (_Float16)(v - (float)a);
  }
}
__device__ __forceinline__ void h2_split(float x, float s, _Float16& a, _Float16& b) {
  const float v = x * s;
  a = (_Float16)v;
  b = (_Float16)(v - (float)a);
}

// max over the wave of a non-negative value, on the vector pipe alone (DPP row shifts + row broadcasts); the result is
// wave-uniform (read from lane 63)
__device__ __forceinline__ float wave_amax(float v) {
#define AFD_DPP_MAX(CTRL, ROWMASK)                                                                                          \
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, true)))
  AFD_DPP_MAX(0x111, 0xf);       // row_shr:1
  AFD_DPP_MAX(0x112, 0xf);       // row_shr:2
  AFD_DPP_MAX(0x114, 0xf);       // row_shr:4
  AFD_DPP_MAX(0x118, 0xf);       // row_shr:8  (lane 15 of every row: the row's max)
  AFD_DPP_MAX(0x142, 0xa);       // row_bcast:15 into rows 1 and 3
  AFD_DPP_MAX(0x143, 0xc);       // row_bcast:31 into rows 2 and 3
#undef AFD_DPP_MAX
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- weight image of the f16x2 direct convolution -----------------------------------------------------------------------
//   records Wp[piece 2][tap 9][k/8][n][8 fp16] (36 bytes per (k, n) pair), then N floats 1 / s_n at byte offset 36 K N
__device__ __forceinline__ long h2_rec(int piece, int tap, int k, int n, int K, int N) {
  return (((long)piece * 9 + tap) * (K >> 3) + (k >> 3)) * N + n;
}
__host__ __device__ inline size_t h2_scale_offset_bytes(int K, int N) { return (size_t)36 * K * N; }

// phase 1: one wave per ROW of an image -- rows = output channels of the forward image (r < Cout) or input channels of the
// dgrad image -- the 64 lanes stride over the row's K x 9 weights; wave `wi` of the layer's `nw` takes rows wi, wi + nw, ..
// and writes 1 / s_n behind the image.
__device__ __forceinline__ void h2_wscale_rows(const float* __restrict__ w, void* Wf, void* Wd, int Cin, int Cout, int wi, int nw, int lane) {
  const int nf = Wf ? Cout : 0, nd = Wd ? Cin : 0;
  for (int r = wi; r < nf + nd; r += nw) {
    const bool fwd = r < nf;
    float m0 = 0.f, m1 = 0.f;
    if (fwd) {
      const float* p = w + (long)r * Cin * 9;
      const int n = Cin * 9;
      int i = lane;
      for (; i + 64 < n; i += 128) { m0 = fmaxf(m0, fabsf(p[i])); m1 = fmaxf(m1, fabsf(p[i + 64])); }
      if (i < n) m0 = fmaxf(m0, fabsf(p[i]));
    } else {
      const int ci = r - nf, n = Cout * 9;
      for (int i = lane; i < n; i += 64) {
        const int co = i / 9, t = i - co * 9;
        m0 = fmaxf(m0, fabsf(w[((long)co * Cin + ci) * 9 + t]));
      }
    }
    const float m = wave_amax(fmaxf(m0, m1));
    if (lane == 0) {
      const float inv = h2_inv_pow2(h2_scale_for(m));
      if (fwd) reinterpret_cast<float*>(static_cast<uint8_t*>(Wf) + h2_scale_offset_bytes(Cin, Cout))[r] = inv;
      else reinterpret_cast<float*>(static_cast<uint8_t*>(Wd) + h2_scale_offset_bytes(Cout, Cin))[r - nf] = inv;
    }
  }
}

// phase 2: a wave = one 8 x 8 block (8 output x 8 input channels), lane = (ci = lane & 7, co = lane >> 3): every store
// instruction writes eight 16-byte records = one contiguous 128-byte run (2 bytes per lane)
__device__ __forceinline__ void h2_weights_block(const float* __restrict__ w, void* Wf, void* Wd, int Cin, int Cout, int blk, int lane) {
  const int nci8 = Cin >> 3;
  if (blk >= nci8 * (Cout >> 3)) return;
  const int ci = (blk % nci8) * 8 + (lane & 7), co = (blk / nci8) * 8 + (lane >> 3);
  const float* p = w + ((long)co * Cin + ci) * 9;
  _Float16* wf = static_cast<_Float16*>(Wf);
  _Float16* wd = static_cast<_Float16*>(Wd);
  const float sf = Wf ? h2_inv_pow2(reinterpret_cast<const float*>(static_cast<const uint8_t*>(Wf) + h2_scale_offset_bytes(Cin, Cout))[co]) : 0.f;
  const float sd = Wd ? h2_inv_pow2(reinterpret_cast<const float*>(static_cast<const uint8_t*>(Wd) + h2_scale_offset_bytes(Cout, Cin))[ci]) : 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float v = p[t];
    _Float16 a, b;
    if (Wf) {                                                                            // forward: n = co, k = ci
      h2_split(v, sf, a, b);
      wf[h2_rec(0, t, ci, co, Cin, Cout) * 8 + (ci & 7)] = a;
      wf[h2_rec(1, t, ci, co, Cin, Cout) * 8 + (ci & 7)] = b;
    }
    if (Wd) {                                                                            // dgrad: n = ci, k = co, tap flipped
      h2_split(v, sd, a, b);
      wd[h2_rec(0, 8 - t, co, ci, Cout, Cin) * 8 + (co & 7)] = a;
      wd[h2_rec(1, 8 - t, co, ci, Cout, Cin) * 8 + (co & 7)] = b;
    }
  }
}

}  // namespace afd

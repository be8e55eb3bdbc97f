// bf3_weights.h -- the exact three-piece bf16 split and the weight image of the direct bf16x3 convolution (bf3.hip);
// shared with wino.hip, whose batched weight-transform launch builds both kinds of image.
#pragma once
#include "common.h"

namespace afd {

using bf8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ void bf3_split(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x; const float r = x - (float)a;
  b = (__bf16)r; const float r2 = r - (float)b;
  c = (__bf16)r2;
}

// ---- weights ---------------------------------------------------------------------------------------------------------
// record index of (piece, tap, k, n) in an image with reduction width K and N output channels
__device__ __forceinline__ long bf3_rec(int piece, int tap, int k, int n, int K, int N) {
  return ((((long)piece * 9 + tap) * (K >> 4) + (k >> 4)) * 2 + ((k >> 3) & 1)) * N + n;
}
// a wave = one 8 x 8 block (8 output x 8 input channels), lane = (ci = lane & 7, co = lane >> 3): every store instruction
// writes eight 16-byte records = one contiguous 128-byte run (2 bytes per lane)
__device__ __forceinline__ void bf3_weights_block(const float* __restrict__ w, __bf16* __restrict__ Wf, __bf16* __restrict__ Wd,
                                                  int Cin, int Cout, int blk, int lane) {
  const int nci8 = Cin >> 3;
  if (blk >= nci8 * (Cout >> 3)) return;
  const int ci = (blk % nci8) * 8 + (lane & 7), co = (blk / nci8) * 8 + (lane >> 3);
  const float* p = w + ((long)co * Cin + ci) * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    __bf16 q[3];
    bf3_split(p[t], q[0], q[1], q[2]);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      if (Wf) Wf[bf3_rec(pc, t, ci, co, Cin, Cout) * 8 + (ci & 7)] = q[pc];           // forward: n = co, k = ci
      if (Wd) Wd[bf3_rec(pc, 8 - t, co, ci, Cout, Cin) * 8 + (co & 7)] = q[pc];       // dgrad:   n = ci, k = co, tap flipped
    }
  }
}

}  // namespace afd

// ends.hip -- F5 at the output end of the network (round 2): the 1x1 output layer (1 or 3 output channels) and its input
// gradient.  A few MFLOP per image against 4 * 32 * H * W bytes: memory-bound streams that the tiled matrix kernels served at
// 1.5 TB/s (24.5 us each at B = 256); as plain vector kernels that read every byte once they run at 3.7 / 4.3 TB/s (10 /
// 8.5 us).  The first layer's weight gradient got the same treatment (wgrad_cin3 in bf3_wgrad.hip: 70 -> 24 us); its FORWARD
// was tried as a vector kernel too (lane = pixel, 27 inputs in registers, wave-uniform weights) and measured 29-39 us against
// the tiled kernel's 25: with one image per workgroup a wave issues 3456 dependent-chain FMAs alone on its SIMD -- not kept.
// Any shape or epilogue outside the conditions below stays on the general kernels.
#include <algorithm>
#include <cstdint>
#include "common.h"

namespace afd {

// ---- 1x1, COUT <= 4 outputs:  y[b][co][p] = bias[co] + sum_ci w[co][ci] x[b][ci][p]; a thread owns 4 consecutive pixels
template <int COUT>
__global__ __launch_bounds__(256) void outc_fwd_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                  float* __restrict__ y, int Cin, long HW4, long total4) {
  extern __shared__ float ws[];                                        // [Cin][COUT]
  for (int i = threadIdx.x; i < Cin * COUT; i += blockDim.x) ws[i] = w[(i % COUT) * Cin + i / COUT];
  __syncthreads();
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW4, p4 = i - b * HW4;
    const float4* xp = reinterpret_cast<const float4*>(x) + b * Cin * HW4 + p4;
    float4 acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) { const float bv = bias ? bias[co] : 0.f; acc[co] = make_float4(bv, bv, bv, bv); }
#pragma unroll 4
    for (int ci = 0; ci < Cin; ++ci) {
      const float4 v = xp[(long)ci * HW4];
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        const float wv = ws[ci * COUT + co];
        acc[co].x = fmaf(wv, v.x, acc[co].x); acc[co].y = fmaf(wv, v.y, acc[co].y);
        acc[co].z = fmaf(wv, v.z, acc[co].z); acc[co].w = fmaf(wv, v.w, acc[co].w);
      }
    }
    float4* yp = reinterpret_cast<float4*>(y) + b * COUT * HW4 + p4;
#pragma unroll
    for (int co = 0; co < COUT; ++co) yp[(long)co * HW4] = acc[co];
  }
}

// ---- its input gradient:  dx[b][ci][p] = sum_co w[co][ci] dy[b][co][p]
template <int COUT>
__global__ __launch_bounds__(256) void outc_dgrad_k(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                    int Cin, long HW4, long total4) {
  extern __shared__ float ws[];                                        // [Cin][COUT]
  for (int i = threadIdx.x; i < Cin * COUT; i += blockDim.x) ws[i] = w[(i % COUT) * Cin + i / COUT];
  __syncthreads();
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW4, p4 = i - b * HW4;
    const float4* gp = reinterpret_cast<const float4*>(dy) + b * COUT * HW4 + p4;
    float4 g[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) g[co] = gp[(long)co * HW4];
    float4* xp = reinterpret_cast<float4*>(dx) + b * Cin * HW4 + p4;
#pragma unroll 4
    for (int ci = 0; ci < Cin; ++ci) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        const float wv = ws[ci * COUT + co];
        a.x = fmaf(wv, g[co].x, a.x); a.y = fmaf(wv, g[co].y, a.y); a.z = fmaf(wv, g[co].z, a.z); a.w = fmaf(wv, g[co].w, a.w);
      }
      xp[(long)ci * HW4] = a;
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static int g_ends_mode = 0;          // afd_debug_conv_path 92 / 93: these kernels by the rule / never
void ends_set_mode(int m) { g_ends_mode = m; }

bool ends_fwd(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int Cin, int Cout, int H, int W,
              int ksize, int act, hipStream_t s) {
  if (g_ends_mode == 1 || res || act) return false;
  const long HW = (long)H * W;
  if (ksize == 1 && Cout <= 4 && Cin <= 1024 && HW % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
    const long total4 = (long)B * HW / 4;
    const unsigned grid = (unsigned)std::min<long>((total4 + 255) / 256, 4096);
    const size_t lds = sizeof(float) * Cin * Cout;
#define AFD_OUTC(C_) hipLaunchKernelGGL(outc_fwd_k<C_>, dim3(grid), dim3(256), lds, s, x, w, bias, y, Cin, HW / 4, total4)
    if (Cout == 1) AFD_OUTC(1); else if (Cout == 2) AFD_OUTC(2); else if (Cout == 3) AFD_OUTC(3); else AFD_OUTC(4);
#undef AFD_OUTC
    return true;
  }
  return false;
}

bool ends_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H, int W, int ksize, hipStream_t s) {
  if (g_ends_mode == 1) return false;
  const long HW = (long)H * W;
  if (ksize == 1 && Cout <= 4 && Cin <= 1024 && HW % 4 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(dx) & 15) == 0) {
    const long total4 = (long)B * HW / 4;
    const unsigned grid = (unsigned)std::min<long>((total4 + 255) / 256, 4096);
    const size_t lds = sizeof(float) * Cin * Cout;
#define AFD_OUTC(C_) hipLaunchKernelGGL(outc_dgrad_k<C_>, dim3(grid), dim3(256), lds, s, dy, w, dx, Cin, HW / 4, total4)
    if (Cout == 1) AFD_OUTC(1); else if (Cout == 2) AFD_OUTC(2); else if (Cout == 3) AFD_OUTC(3); else AFD_OUTC(4);
#undef AFD_OUTC
    return true;
  }
  return false;
}

}  // namespace afd

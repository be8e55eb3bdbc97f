// common.h -- shared helpers for the gfx950 kernels of libafd_hip.so (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/afd.h"

namespace afd {

int set_error(int code, const char* fmt, ...);   // host.cpp

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(AFD_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return AFD_OK;
}

#define AFD_REQUIRE(cond, ...) \
  do { if (!(cond)) return afd::set_error(AFD_EINVAL, __VA_ARGS__); } while (0)

inline hipStream_t as_stream(afd_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Kernels that need more than 64 KB of dynamic LDS opt in through hipFuncSetAttribute -- once per (DEVICE, kernel): the
// attribute belongs to the device's copy of the function, so a process that drives several GPUs must set it on each.
int lds_opt_in_impl(const void* kern, size_t lds);   // host.cpp
template <class K> inline int lds_opt_in(K kern, size_t lds) {
  return lds <= 64 * 1024 ? AFD_OK : lds_opt_in_impl(reinterpret_cast<const void*>(kern), lds);
}

constexpr int kWave = 64;                  // CDNA wavefront
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;

// exact (erf) GELU and its derivative -- nn.GELU(approximate='none').
// erf by Abramowitz-Stegun 7.1.26: erf(z) = 1 - (a1 t + ... + a5 t^5) e^{-z^2}, t = 1/(1 + p z), z >= 0,
// |error| <= 1.5e-7 analytically (6e-7 in fp32 arithmetic; GELU rel-L2 2e-8 against fp64 erf).  One rcp and
// one exp instead of the library erff's ~25-instruction two-branch form; with z = u/sqrt(2) the same
// exponential e^{-u^2/2} is the Gaussian of the derivative, so gelu' costs no second transcendental.
__device__ __forceinline__ void erf_gauss(float u, float& erf_v, float& gauss) {
  const float z = fabsf(u) * kInvSqrt2;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  gauss = __expf(-z * z);                                   // = exp(-u^2 / 2)
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = fmaf(-(p * t), gauss, 1.0f);
  erf_v = copysignf(e, u);
}
__device__ __forceinline__ float gelu_erf(float u) {
  float e, g; erf_gauss(u, e, g);
  return 0.5f * u * (1.0f + e);
}
__device__ __forceinline__ float gelu_erf_grad(float u) {
  float e, g; erf_gauss(u, e, g);
  return fmaf(u * kInvSqrt2Pi, g, 0.5f * (1.0f + e));
}

// neighbour exchange inside a wave (lane-1 / lane+1); callers mask the plane borders
__device__ __forceinline__ float lane_left(float v) { return __shfl_up(v, 1, kWave); }
__device__ __forceinline__ float lane_right(float v) { return __shfl_down(v, 1, kWave); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

// the wave's sum in lane 63 only, on the vector pipe alone (DPP row shifts + row broadcasts, 6 instructions, no LDS
// round trips): for kernels that reduce MANY registers at once (wave_sum's cross-half steps are ds_bpermute)
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
#define AFD_DPP_ADD(CTRL, ROWMASK)                                                                                          \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, true))
  AFD_DPP_ADD(0x111, 0xf);       // row_shr:1
  AFD_DPP_ADD(0x112, 0xf);       // row_shr:2
  AFD_DPP_ADD(0x114, 0xf);       // row_shr:4  (lane i of a row now holds the sum of lanes max(0, i-7) .. i)
  AFD_DPP_ADD(0x118, 0xf);       // row_shr:8  (lane 15 of every row: the row's sum)
  AFD_DPP_ADD(0x142, 0xa);       // row_bcast:15 into rows 1 and 3
  AFD_DPP_ADD(0x143, 0xc);       // row_bcast:31 into rows 2 and 3
#undef AFD_DPP_ADD
  return v;
}

// block-wide sum for blockDim.x <= 1024 (all threads get the result); `red` >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];   // fixed order: deterministic
  return t;
}

struct Taps {                               // filter taps travel as kernel arguments (no H2D copy)
  float k[AFD_MAX_TAPS * AFD_MAX_TAPS];
};
struct Taps3 { float k[9]; };

}  // namespace afd

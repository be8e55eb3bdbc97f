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

constexpr int kWave = 64;                  // CDNA wavefront
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;

// exact (erf) GELU and its derivative -- nn.GELU(approximate='none')
__device__ __forceinline__ float gelu_erf(float u) { return 0.5f * u * (1.0f + erff(u * kInvSqrt2)); }
__device__ __forceinline__ float gelu_erf_grad(float u) {
  return 0.5f * (1.0f + erff(u * kInvSqrt2)) + u * kInvSqrt2Pi * __expf(-0.5f * u * u);
}

// neighbour exchange inside a wave (lane-1 / lane+1); callers mask the plane borders
__device__ __forceinline__ float lane_left(float v) { return __shfl_up(v, 1, kWave); }
__device__ __forceinline__ float lane_right(float v) { return __shfl_down(v, 1, kWave); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
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

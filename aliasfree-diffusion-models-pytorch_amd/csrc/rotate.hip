// rotate.hip -- F17 (Config E): scipy.ndimage.rotate(x, angle, axes=(2,3), reshape=False, order=3,
// mode='grid-wrap', prefilter=True) on the device, replacing the reference's per-step
// D2H -> CPU spline -> H2D round trip (ddpm_models.py:421-429).
//
// scipy works plane by plane in float64: (1) cubic B-spline prefilter along axis 0 then axis 1 -- gain
// (1-z)(1-1/z) = 6, pole z = sqrt(3)-2, periodic ('grid-wrap') initial conditions, causal then
// anti-causal recursion; (2) for every output pixel the input coordinate  M (o) + offset  is mapped
// into the period, the 4x4 footprint starts at floor(c)-1, indices wrap, weights are the cubic
// B-spline pieces.  The recursions and weight formulas below restate that arithmetic in the same
// order and precision (fp64); the result is rounded once to fp32 like scipy's float32 output array.
#include "common.h"

namespace afd {

constexpr double kPole = -0.26794919243112270647;      // sqrt(3) - 2

// one thread per line; `stride` = element distance along the filtered axis, lines are `lstride` apart
__global__ __launch_bounds__(128) void spline3_prefilter_wrap(double* __restrict__ c, int n, long stride, long lines,
                                                              int inner, long lstride_outer, long lstride_inner) {
  const long ln = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ln >= lines) return;
  double* p = c + (ln / inner) * lstride_outer + (ln % inner) * lstride_inner;
  const double z = kPole;
  if (n < 2) return;
  const double gain = (1.0 - z) * (1.0 - 1.0 / z);
  for (int i = 0; i < n; ++i) p[i * stride] *= gain;
  // causal initialisation (periodic): c[0] += sum_{i=1}^{n-1} z^i c[n-i];  c[0] /= 1 - z^n
  double zi = z, c0 = p[0];
  for (int i = 1; i < n; ++i) { c0 += zi * p[(long)(n - i) * stride]; zi *= z; }
  p[0] = c0 / (1.0 - zi);
  for (int i = 1; i < n; ++i) p[i * stride] += z * p[(i - 1) * stride];
  // anti-causal initialisation: c[n-1] += sum_{i=0}^{n-2} z^(i+1) c[i];  c[n-1] *= z / (z^n - 1)
  zi = z; double cl = p[(long)(n - 1) * stride];
  for (int i = 0; i < n - 1; ++i) { cl += zi * p[i * stride]; zi *= z; }
  p[(long)(n - 1) * stride] = cl * (z / (zi - 1.0));
  for (int i = n - 2; i >= 0; --i) p[i * stride] = z * (p[(i + 1) * stride] - p[i * stride]);
}

__global__ void f32_to_f64(const float* __restrict__ x, double* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (double)x[i];
}

__device__ __forceinline__ double map_grid_wrap(double in, int len) {
  if (len <= 1) return 0.0;
  if (in < 0) in += (double)len * (double)((long)((-1.0 - in) / len) + 1);
  else if (in > len - 1) { in -= (double)len * (double)((long)((in + 1.0) / len)); if (in < 0) in += len; }
  return in;
}
__device__ __forceinline__ void spline3_weights(double x, double (&w)[4]) {
  x -= floor(x);
  const double y = x, z = 1.0 - x;
  w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
  w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
  w[0] = z * z * z / 6.0;
  w[3] = 1.0 - w[0] - w[1] - w[2];
}
__device__ __forceinline__ int wrap_idx(int i, int n) { i %= n; return i < 0 ? i + n : i; }

// out[p, oy, ox] = sum_{i,j} coef[p, wrap(sy+i), wrap(sx+j)] * wy[i] * wx[j]
__global__ void spline3_affine_wrap(const double* __restrict__ coef, float* __restrict__ out, long planes, int H, int W,
                                    double m00, double m01, double m10, double m11, double off0, double off1) {
  const long total = planes * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = i % W, oy = (i / W) % H; const long p = i / ((long)W * H);
    double cy = m00 * oy + m01 * ox + off0;
    double cx = m10 * oy + m11 * ox + off1;
    cy = map_grid_wrap(cy, H); cx = map_grid_wrap(cx, W);
    double wy[4], wx[4];
    spline3_weights(cy, wy); spline3_weights(cx, wx);
    const int sy = (int)floor(cy) - 1, sx = (int)floor(cx) - 1;
    const double* cp = coef + p * (long)H * W;
    double t = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = wrap_idx(sy + a, H);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        double v = cp[(long)yy * W + wrap_idx(sx + b, W)];
        v *= wy[a]; v *= wx[b];
        t += v;
      }
    }
    out[i] = (float)t;
  }
}

}  // namespace afd
using namespace afd;

extern "C" {

size_t afd_rotate_workspace_bytes(long planes, int H, int W) { return sizeof(double) * (size_t)planes * H * W; }

// matrix / offset are the affine map scipy builds for `rotate`: in = M @ out + offset (row, col order)
int afd_affine_spline3_wrap(const float* x, float* y, long planes, int H, int W, const double* matrix4, const double* offset2,
                            void* workspace, afd_stream_t st) {
  AFD_REQUIRE(x && y && matrix4 && offset2 && workspace && planes > 0 && H > 0 && W > 0, "afd_affine_spline3_wrap: bad argument");
  hipStream_t s = as_stream(st);
  double* c = static_cast<double*>(workspace);
  const long n = planes * H * W;
  long g = (n + 255) / 256; if (g > 32768) g = 32768;
  hipLaunchKernelGGL(f32_to_f64, dim3((unsigned)g), dim3(256), 0, s, x, c, n);
  // axis 0 (rows direction): lines = (plane, column); then axis 1: lines = (plane, row)
  long lines = planes * W;
  hipLaunchKernelGGL(spline3_prefilter_wrap, dim3((unsigned)((lines + 127) / 128)), dim3(128), 0, s, c, H, (long)W, lines, W, (long)H * W, 1L);
  lines = planes * H;
  hipLaunchKernelGGL(spline3_prefilter_wrap, dim3((unsigned)((lines + 127) / 128)), dim3(128), 0, s, c, W, 1L, lines, H, (long)H * W, (long)W);
  hipLaunchKernelGGL(spline3_affine_wrap, dim3((unsigned)g), dim3(256), 0, s, c, y, planes, H, W,
                     matrix4[0], matrix4[1], matrix4[2], matrix4[3], offset2[0], offset2[1]);
  return check_launch("afd_affine_spline3_wrap");
}

}  // extern "C"

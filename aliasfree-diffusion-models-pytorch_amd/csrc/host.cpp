// host.cpp -- host-side entry points of libafd_hip.so: errors, version, F1 filter design.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <utility>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "../../include/afd.h"

namespace afd {
static thread_local char g_err[512] = "";
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
// (device, kernel) pairs whose dynamic-LDS limit has been raised (common.h: lds_opt_in)
int lds_opt_in_impl(const void* kern, size_t lds) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({dev, kern})) return AFD_OK;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return set_error(AFD_ELAUNCH, "hipFuncSetAttribute(%zu B of LDS): %s", lds, hipGetErrorString(e));
  done.insert({dev, kern});
  return AFD_OK;
}
}  // namespace afd

extern "C" {

const char* afd_version(void) { return "afd-hip 0.1 (gfx950)"; }
const char* afd_last_error(void) { return afd::g_err; }

int afd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// F1 -- filtrs.py:20-37.  k[x,y] = w_c J1(w_c r) / (2 pi r), r = distance to ((N-1)/2,(N-1)/2);
// odd N: centre tap = w_c^2/(4 pi) (the r -> 0 limit) set before windowing; optional Kaiser
// window w[n] = I0(beta sqrt(1-((n-a)/a)^2)) / I0(beta), a=(N-1)/2 (numpy.kaiser), outer product;
// divide by the fp64 sum (numpy pairwise order for N*N <= 128 is a plain left-to-right... see below);
// round once to fp32.
int afd_lowpass_kernel(double omega_c, int N, int has_beta, double beta, float* taps_out) {
  if (!taps_out) return afd::set_error(AFD_EINVAL, "afd_lowpass_kernel: taps_out is NULL");
  if (N < 1 || N > AFD_MAX_TAPS) return afd::set_error(AFD_EINVAL, "afd_lowpass_kernel: N=%d outside [1,%d]", N, AFD_MAX_TAPS);
  const double c = (N - 1) / 2.0;
  std::vector<double> k((size_t)N * N), win(N, 1.0);
  for (int x = 0; x < N; ++x)
    for (int y = 0; y < N; ++y) {
      const double dx = x - c, dy = y - c;
      const double r = std::sqrt(dx * dx + dy * dy);
      k[(size_t)x * N + y] = (r == 0.0) ? omega_c * omega_c / (4.0 * M_PI)
                                         : omega_c * std::cyl_bessel_j(1.0, omega_c * r) / (2.0 * M_PI * r);
    }
  if (has_beta) {
    if (N == 1) win[0] = 1.0;
    else {
      const double i0b = std::cyl_bessel_i(0.0, std::fabs(beta));
      for (int n = 0; n < N; ++n) {
        const double z = (n - c) / c;
        win[n] = std::cyl_bessel_i(0.0, std::fabs(beta) * std::sqrt(std::fmax(0.0, 1.0 - z * z))) / i0b;
      }
    }
    for (int x = 0; x < N; ++x)
      for (int y = 0; y < N; ++y) k[(size_t)x * N + y] *= win[x] * win[y];
  }
  // numpy.sum on a contiguous fp64 array: pairwise summation with an 8-way unrolled base case for
  // blocks < 128 elements; restate that order so the normaliser rounds identically.
  auto np_sum = [&](const double* a, size_t n) -> double {
    struct L { static double run(const double* a, size_t n) {
      if (n < 8) { double s = 0.0; for (size_t i = 0; i < n; ++i) s += a[i]; return s; }
      if (n <= 128) {
        double r[8]; for (int j = 0; j < 8; ++j) r[j] = a[j];
        size_t i = 8;
        for (; i + 8 <= n; i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) s += a[i];
        return s;
      }
      size_t n2 = n / 2; n2 -= n2 % 8;
      return run(a, n2) + run(a + n2, n - n2);
    } };
    return L::run(a, n);
  };
  const double s = np_sum(k.data(), k.size());
  for (size_t i = 0; i < k.size(); ++i) taps_out[i] = (float)(k[i] / s);
  return AFD_OK;
}

}  // extern "C"

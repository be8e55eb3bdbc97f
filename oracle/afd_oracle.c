/* afd_oracle.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * Plain-C restatement of the byte/element-exact parts of the reference's hot path, independent of
 * torch / scipy, used by tests/ to cross-check both the Python oracle (oracle/ref_ops.py) and the HIP
 * kernels:
 *   F1  circularLowpassKernel      modules/filtrs.py:20-37   (J1 and I0 from Bessel's integrals)
 *   F2  custom_upsample            modules/filtrs.py:79-94
 *   F3  custom_downsample          modules/filtrs.py:71-77
 *   F4  up -> erf-GELU -> down     modules/ddpm_utils.py:123-125
 *   F6  GroupNorm(1,C)             modules/ddpm_utils.py:113
 *   F12 noise schedule             modules/ddpm_models.py:309-315
 *   F14 noise_images, F16 denoise update + uint8 quantisation  modules/ddpm_models.py:317-321,374,381-382
 * Pinned by tests/golden/*.npz (vectors produced by running the reference): tests/test_c_oracle.py.
 * Build: make -C oracle  ->  oracle/libafd_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PI 3.14159265358979323846

/* J1(x) = (1/pi) int_0^pi cos(tau - x sin tau) dtau ; I0(x) = (1/pi) int_0^pi exp(x cos tau) dtau.
 * Periodic analytic integrands: the trapezoid rule converges geometrically. */
static double bessel_j1(double x) {
  const int M = 512; double s = 0.0;
  for (int i = 0; i < M; ++i) { double t = PI * (i + 0.5) / M; s += cos(t - x * sin(t)); }
  return s / M;
}
static double bessel_i0(double x) {
  const int M = 512; double s = 0.0;
  for (int i = 0; i < M; ++i) { double t = PI * (i + 0.5) / M; s += exp(x * cos(t)); }
  return s / M;
}

void orc_lowpass_kernel(double omega_c, int N, int has_beta, double beta, float* out) {
  double* k = (double*)malloc(sizeof(double) * N * N);
  double c = (N - 1) / 2.0, sum = 0.0;
  for (int x = 0; x < N; ++x)
    for (int y = 0; y < N; ++y) {
      double r = sqrt((x - c) * (x - c) + (y - c) * (y - c));
      k[x * N + y] = (r == 0.0) ? omega_c * omega_c / (4 * PI) : omega_c * bessel_j1(omega_c * r) / (2 * PI * r);
    }
  if (has_beta && N > 1) {
    double i0b = bessel_i0(fabs(beta));
    for (int x = 0; x < N; ++x)
      for (int y = 0; y < N; ++y) {
        double zx = (x - c) / c, zy = (y - c) / c;
        double wx = bessel_i0(fabs(beta) * sqrt(fmax(0.0, 1 - zx * zx))) / i0b;
        double wy = bessel_i0(fabs(beta) * sqrt(fmax(0.0, 1 - zy * zy))) / i0b;
        k[x * N + y] *= wx * wy;
      }
  }
  for (int i = 0; i < N * N; ++i) sum += k[i];
  for (int i = 0; i < N * N; ++i) out[i] = (float)(k[i] / sum);
  free(k);
}

/* zero 'same' padding: lo = (N-1)/2 taps before, the rest after (torch conv2d padding='same') */
void orc_up2(const float* x, float* y, int planes, int H, int W, const float* k, int N) {
  int lo = (N - 1) / 2, H2 = 2 * H, W2 = 2 * W;
  for (int p = 0; p < planes; ++p)
    for (int r = 0; r < H2; ++r)
      for (int q = 0; q < W2; ++q) {
        double acc = 0.0;
        for (int a = 0; a < N; ++a) {
          int zr = r + a - lo; if (zr < 0 || zr >= H2 || (zr & 1)) continue;
          for (int b = 0; b < N; ++b) {
            int zc = q + b - lo; if (zc < 0 || zc >= W2 || (zc & 1)) continue;
            acc += (double)k[a * N + b] * x[((size_t)p * H + zr / 2) * W + zc / 2];
          }
        }
        y[((size_t)p * H2 + r) * W2 + q] = (float)acc;
      }
}

void orc_down2(const float* x, float* y, int planes, int H, int W, const float* k, int N) {
  int lo = (N - 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  for (int p = 0; p < planes; ++p)
    for (int r = 0; r < Ho; ++r)
      for (int q = 0; q < Wo; ++q) {
        double acc = 0.0;
        for (int a = 0; a < N; ++a) {
          int xr = 2 * r + a - lo; if (xr < 0 || xr >= H) continue;
          for (int b = 0; b < N; ++b) {
            int xc = 2 * q + b - lo; if (xc < 0 || xc >= W) continue;
            acc += (double)k[a * N + b] * x[((size_t)p * H + xr) * W + xc];
          }
        }
        y[((size_t)p * Ho + r) * Wo + q] = (float)acc;
      }
}

static float gelu_erf(float u) { return (float)(0.5 * (double)u * (1.0 + erf((double)u / sqrt(2.0)))); }

void orc_filt_act(const float* x, float* y, int planes, int H, int W, const float* ku, const float* kd, int N) {
  size_t n2 = (size_t)planes * 4 * H * W;
  float* u = (float*)malloc(sizeof(float) * n2);
  orc_up2(x, u, planes, H, W, ku, N);
  for (size_t i = 0; i < n2; ++i) u[i] = gelu_erf(u[i]);
  orc_down2(u, y, planes, 2 * H, 2 * W, kd, N);
  free(u);
}

void orc_groupnorm1(const float* x, float* y, int B, int C, int HW, float eps, const float* gamma, const float* beta) {
  size_t n = (size_t)C * HW;
  for (int b = 0; b < B; ++b) {
    const float* xp = x + b * n; double m = 0, v = 0;
    for (size_t i = 0; i < n; ++i) m += xp[i];
    m /= n;
    for (size_t i = 0; i < n; ++i) v += (xp[i] - m) * (xp[i] - m);
    double rstd = 1.0 / sqrt(v / n + eps);
    for (size_t i = 0; i < n; ++i) y[b * n + i] = (float)((xp[i] - m) * rstd * gamma[i / HW] + beta[i / HW]);
  }
}

/* alpha = 1 - beta (fp32), alpha_hat = running product kept in DOUBLE and rounded to fp32 per element:
 * ATen's CPU cumprod accumulates float inputs in acc_type<float> = double (found by pinning against the
 * reference's tables; a plain fp32 running product differs from them in the last bit from index 4 on).
 * beta itself comes from torch.linspace on the host in both the reference and the product: its
 * vectorised evaluation order is an ATen implementation detail that is not restated here. */
void orc_schedule_from_beta(int T, const float* beta, float* alpha, float* alpha_hat) {
  double acc = 1.0;
  for (int i = 0; i < T; ++i) { alpha[i] = 1.0f - beta[i]; acc = acc * (double)alpha[i]; alpha_hat[i] = (float)acc; }
}

void orc_noise_images(const float* x, const float* eps, const int64_t* t, const float* alpha_hat, float* xt, int B, long per) {
  for (int b = 0; b < B; ++b) {
    float ah = alpha_hat[t[b]];
    volatile float sa = sqrtf(ah), sb = sqrtf(1.0f - ah);
    for (long i = 0; i < per; ++i) {
      volatile float l = sa * x[b * per + i], r = sb * eps[b * per + i];
      xt[b * per + i] = l + r;
    }
  }
}

void orc_denoise_step(const float* x, const float* eps, const float* noise, const float* alpha, const float* alpha_hat,
                      const float* beta, int i, float* out, long n) {
  volatile float c1 = 1.0f / sqrtf(alpha[i]);
  volatile float c2 = (1.0f - alpha[i]) / sqrtf(1.0f - alpha_hat[i]);
  volatile float sb = sqrtf(beta[i]);
  for (long j = 0; j < n; ++j) {
    volatile float pe = c2 * eps[j];
    volatile float inner = x[j] - pe;
    volatile float lhs = c1 * inner;
    volatile float nz = noise ? sb * noise[j] : 0.0f;
    out[j] = lhs + nz;
  }
}

void orc_quantize_u8(const float* x, uint8_t* out, long n) {
  for (long j = 0; j < n; ++j) {
    float v = x[j]; v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
    volatile float s = (v + 1.0f) / 2.0f;
    volatile float q = s * 255.0f;
    out[j] = (uint8_t)(int)q;
  }
}

// ddpm.hip -- F14/F16 DDPM noise / denoise / quantise, MSE loss, fused AdamW.
//
// noise_images / denoise_step / quantize restate the reference's fp32 expression ORDER with one
// IEEE rounding per torch op (no FMA contraction, correctly rounded sqrt and divide), so given the
// same inputs they are bit-identical to the reference's CPU path (ddpm_models.py:317-321,367-374,381-385).
// hipcc keeps `/` and sqrtf correctly rounded by default; `#pragma clang fp contract(off)` below
// stops a*b+c from fusing.  (The __f*_rn intrinsics are NOT used: without
// OCML_BASIC_ROUNDED_OPERATIONS this toolchain maps __fsqrt_rn to the approximate native sqrt.)
#include "common.h"

#pragma clang fp contract(off)

namespace afd {

static inline int gs_grid(long total, int block = 256) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 32768 ? 32768 : g));
}
#define AFD_GRID_STRIDE(i, total) \
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (total); i += (long)gridDim.x * blockDim.x)

// x_t = sqrt(ah[t]) * x + sqrt(1 - ah[t]) * eps
__global__ void noise_images_k(const float* __restrict__ x, const float* __restrict__ eps, const int64_t* __restrict__ t,
                               const float* __restrict__ alpha_hat, float* __restrict__ xt, long per, long total) {
  AFD_GRID_STRIDE(i, total) {
    const long b = i / per;
    const float ah = alpha_hat[t[b]];
    const float sa = sqrtf(ah);
    const float sb = sqrtf(1.0f - ah);
    const float l = sa * x[i], r = sb * eps[i];
    xt[i] = l + r;
  }
}

// x' = 1/sqrt(a) * (x - ((1-a)/sqrt(1-ah)) * eps) + sqrt(b) * noise
__global__ void denoise_step_k(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ noise,
                               const float* __restrict__ alpha, const float* __restrict__ alpha_hat, const float* __restrict__ beta,
                               int step_arg, const int64_t* __restrict__ step_dev, float* __restrict__ out, long n) {
  const int step = step_dev ? (int)step_dev[0] : step_arg;     // device-resident index: a captured graph replays for every i
  const float a = alpha[step], ah = alpha_hat[step], bt = beta[step];
  const float c1 = 1.0f / sqrtf(a);
  const float c2 = (1.0f - a) / sqrtf(1.0f - ah);
  const float sb = sqrtf(bt);
  AFD_GRID_STRIDE(i, n) {
    const float pe = c2 * eps[i];
    const float inner = x[i] - pe;
    const float lhs = c1 * inner;
    const float nz = noise ? sb * noise[i] : 0.0f;       // sqrt(beta) * zeros == +0
    out[i] = lhs + nz;
  }
}

// ((clamp(x,-1,1) + 1) / 2 * 255).type(uint8): truncation toward zero
__global__ void quantize_u8_k(const float* __restrict__ x, uint8_t* __restrict__ out, long n) {
  AFD_GRID_STRIDE(i, n) {
    float v = x[i];
    v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);            // NaN passes through like torch.clamp
    v = ((v + 1.0f) / 2.0f) * 255.0f;
    out[i] = (uint8_t)(int)v;
  }
}

// ---- MSE ------------------------------------------------------------------------------------
constexpr int kMseBlocks = 1024;
__global__ void mse_partial_k(const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ part, long n) {
  __shared__ float red[16];
  float s = 0.f;
  AFD_GRID_STRIDE(i, n) { const float d = p[i] - t[i]; s += d * d; }
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void mse_final_k(const float* __restrict__ part, float* __restrict__ loss, int nparts, float inv_n) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = s * inv_n;
}
__global__ void mse_bwd_k(const float* __restrict__ p, const float* __restrict__ t, const float* __restrict__ dloss,
                          float* __restrict__ dp, long n, float two_over_n) {
  const float g = dloss[0] * two_over_n;
  AFD_GRID_STRIDE(i, n) dp[i] = (p[i] - t[i]) * g;
}

// ---- AdamW (torch.optim.AdamW semantics, decoupled weight decay) ---------------------------
__global__ void adamw_tick_k(float* state, float b1, float b2) {
  // state = {step, 1 - b1^step, 1 - b2^step, unused}; double keeps the powers exact enough for 1e6 steps
  const double step = (double)state[0] + 1.0;
  state[0] = (float)step;
  state[1] = (float)(1.0 - pow((double)b1, step));
  state[2] = (float)(1.0 - pow((double)b2, step));
}
__global__ void adamw_step_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, const float* __restrict__ state, float lr, float b1, float b2, float eps, float wd, float gscale) {
  const float bc1 = state[1], bc2 = state[2];
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2), decay = 1.0f - lr * wd;
  AFD_GRID_STRIDE(i, n) {
    const float gi = g[i] * gscale;
    const float pi = p[i] * decay;
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);            // lerp, as torch does
    const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = pi - step_size * (mi / denom);
    m[i] = mi; v[i] = vi;
  }
}

}  // namespace afd
using namespace afd;

extern "C" {

int afd_noise_images(const float* x, const float* eps, const int64_t* t, const float* alpha_hat, float* x_t,
                     int B, long per_sample, afd_stream_t st) {
  AFD_REQUIRE(x && eps && t && alpha_hat && x_t && B > 0 && per_sample > 0, "afd_noise_images: bad argument");
  const long total = (long)B * per_sample;
  hipLaunchKernelGGL(noise_images_k, dim3(gs_grid(total)), dim3(256), 0, as_stream(st), x, eps, t, alpha_hat, x_t, per_sample, total);
  return check_launch("afd_noise_images");
}
int afd_denoise_step(const float* x, const float* eps_pred, const float* noise, const float* alpha, const float* alpha_hat,
                     const float* beta, int i, float* x_out, long n, afd_stream_t st) {
  AFD_REQUIRE(x && eps_pred && alpha && alpha_hat && beta && x_out && n > 0 && i >= 0, "afd_denoise_step: bad argument");
  hipLaunchKernelGGL(denoise_step_k, dim3(gs_grid(n)), dim3(256), 0, as_stream(st), x, eps_pred, noise, alpha, alpha_hat, beta, i, (const int64_t*)nullptr, x_out, n);
  return check_launch("afd_denoise_step");
}
int afd_denoise_step_dev(const float* x, const float* eps_pred, const float* noise, const float* alpha, const float* alpha_hat,
                         const float* beta, const int64_t* t_dev, float* x_out, long n, afd_stream_t st) {
  AFD_REQUIRE(x && eps_pred && alpha && alpha_hat && beta && t_dev && x_out && n > 0, "afd_denoise_step_dev: bad argument");
  hipLaunchKernelGGL(denoise_step_k, dim3(gs_grid(n)), dim3(256), 0, as_stream(st), x, eps_pred, noise, alpha, alpha_hat, beta, 0, t_dev, x_out, n);
  return check_launch("afd_denoise_step_dev");
}
int afd_quantize_u8(const float* x, uint8_t* out, long n, afd_stream_t st) {
  AFD_REQUIRE(x && out && n > 0, "afd_quantize_u8: bad argument");
  hipLaunchKernelGGL(quantize_u8_k, dim3(gs_grid(n)), dim3(256), 0, as_stream(st), x, out, n);
  return check_launch("afd_quantize_u8");
}
int afd_mse_fwd(const float* pred, const float* target, float* loss_out, float* workspace, long n, afd_stream_t st) {
  AFD_REQUIRE(pred && target && loss_out && workspace && n > 0, "afd_mse_fwd: bad argument");
  const int nb = gs_grid(n) < kMseBlocks ? gs_grid(n) : kMseBlocks;
  hipLaunchKernelGGL(mse_partial_k, dim3(nb), dim3(256), 0, as_stream(st), pred, target, workspace, n);
  hipLaunchKernelGGL(mse_final_k, dim3(1), dim3(256), 0, as_stream(st), workspace, loss_out, nb, 1.0f / (float)n);
  return check_launch("afd_mse_fwd");
}
int afd_mse_bwd(const float* pred, const float* target, const float* dloss, float* dpred, long n, afd_stream_t st) {
  AFD_REQUIRE(pred && target && dloss && dpred && n > 0, "afd_mse_bwd: bad argument");
  hipLaunchKernelGGL(mse_bwd_k, dim3(gs_grid(n)), dim3(256), 0, as_stream(st), pred, target, dloss, dpred, n, 2.0f / (float)n);
  return check_launch("afd_mse_bwd");
}
int afd_adamw_tick(float* state, float b1, float b2, afd_stream_t st) {
  AFD_REQUIRE(state, "afd_adamw_tick: state is NULL");
  hipLaunchKernelGGL(adamw_tick_k, dim3(1), dim3(1), 0, as_stream(st), state, b1, b2);
  return check_launch("afd_adamw_tick");
}
int afd_adamw_step(float* p, const float* g, float* m, float* v, long n, const float* state,
                   float lr, float b1, float b2, float eps, float wd, float gscale, afd_stream_t st) {
  AFD_REQUIRE(p && g && m && v && state && n > 0, "afd_adamw_step: bad argument");
  hipLaunchKernelGGL(adamw_step_k, dim3(gs_grid(n)), dim3(256), 0, as_stream(st), p, g, m, v, n, state, lr, b1, b2, eps, wd, gscale);
  return check_launch("afd_adamw_step");
}

}  // extern "C"

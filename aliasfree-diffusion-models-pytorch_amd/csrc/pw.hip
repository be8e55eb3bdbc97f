// pw.hip -- pointwise (1x1) convolution forward / dgrad as a STREAMING kernel: the token projections of the attention
// blocks (in_proj, out_proj, the two feed-forward layers; SURVEY F10, ddpm_utils.py:54-74).
//
//   out[b][n][p] = sum_k Wm[n][k] * in[b][k][p]  (+ bias[n]) (GELU) (+ res)      Wm = w (forward) or w^T (dgrad)
//
// These layers have K, N <= 384 and up to 262,144 pixels: 20-100 flop per byte, at or under the fp32 MFMA ridge, so
// the job is to stream `in` and `out` once at HBM rate.  The general conv kernel (one LDS tile per workgroup, a
// single K chunk, hence no pipelining) ran them at 0.3-2.2 TB/s.  Here:
//   * the weights (and bias) of the workgroup's output channels are staged into LDS ONCE ([n][K+1], odd stride);
//   * a WAVE owns 32 consecutive pixels at a time and loops over pixel tiles (persistent grid): the activation
//     fragments go global -> registers directly in the MFMA B layout (lane = pixel, k = 2s + lane/32): per k one
//     128-B run per half-wave, no LDS, no barrier in the loop;
//   * the next tile's fragments are loaded while the current tile's MFMAs run;
//   * accumulator rows are output channels, columns are pixels: every store is a 128-B run per half-wave.
// K <= 64 stays resident in registers (K/2 fragment registers per tile, twice for the prefetch); the wave loops over the
// 32-channel output blocks.  Measured against the tiled kernel (tools/pw_bench.py, B=256): 32->96 @32x32 forward
// 96 -> 41 us (3.3 TB/s), 32->32 @32x32 30 -> 19 us (3.6 TB/s); it loses below ~32k pixels (too few 32-pixel tiles
// to fill the chip) and for K > 64 (register pressure), where the tiled kernel keeps the layer.
#include "common.h"

namespace afd {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int KH>
__global__ __launch_bounds__(256) void pw_mfma(const float* __restrict__ in, const float* __restrict__ w,
                                               const float* __restrict__ bias, const float* __restrict__ res,
                                               float* __restrict__ out, int B, int K, int N, int P, int act, int dgrad,
                                               int n_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int WS = K + 1;                               // K is even -> odd row stride, conflict-free fragment reads
  float* Ws = smem;                                   // [n_per_wg][WS]
  float* bs = smem + n_per_wg * WS;                   // [n_per_wg]
  const int nbase = blockIdx.y * n_per_wg;
  const int nhere = min(n_per_wg, N - nbase);         // live output channels of this workgroup
  for (int i = threadIdx.x; i < n_per_wg * K; i += 256) {
    int n, k;
    if (!dgrad) { n = i / K; k = i - n * K; } else { k = i / n_per_wg; n = i - k * n_per_wg; }
    const int nc = min(nbase + n, N - 1);             // rows past N repeat row N-1; they are never stored
    Ws[n * WS + k] = dgrad ? w[(long)k * N + nc] : w[(long)nc * K + k];
  }
  for (int i = threadIdx.x; i < n_per_wg; i += 256) bs[i] = (bias && nbase + i < N) ? bias[nbase + i] : 0.f;
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const unsigned total = (unsigned)B * (unsigned)P;
  const int ntile = (int)((total + 31u) / 32u);
  const int tstride = gridDim.x * 4;
  const int nblk = (nhere + 31) / 32;
  const int a_lane = l31 * WS + half;

  auto tile_origin = [&](int tile, unsigned& ioff, unsigned& ooff, bool& live) {
    const unsigned f = (unsigned)tile * 32u + (unsigned)l31;
    live = f < total;
    const unsigned fc = live ? f : 0u;
    const unsigned bb = fc / (unsigned)P, pp = fc - bb * (unsigned)P;
    ioff = bb * (unsigned)K * (unsigned)P + pp;
    ooff = bb * (unsigned)N * (unsigned)P + pp;
  };
  auto epilogue = [&](const f32x16& acc, int nt, unsigned ooff, bool live) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nl = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (live && nl < nhere) {
        const unsigned o = ooff + (unsigned)(nbase + nl) * (unsigned)P;
        float v = acc[r] + bs[nl];
        if (act == 1) v = gelu_erf(v);
        if (res) v += res[o];
        out[o] = v;
      }
    }
  };

  // ---- all K/2 fragment registers of a tile resident; loop over 32-channel blocks
  float bc[KH], bn[KH];
  unsigned io = 0, oo = 0, io_n = 0, oo_n = 0; bool lv = false, lv_n = false;
  int t = blockIdx.x * 4 + wv;
  if (t < ntile) {
    tile_origin(t, io, oo, lv);
#pragma unroll
    for (int s = 0; s < KH; ++s) bc[s] = (2 * s < K) ? in[io + (unsigned)(2 * s + half) * (unsigned)P] : 0.f;
  }
  for (; t < ntile; t += tstride) {
    const int tn = t + tstride;
    if (tn < ntile) {
      tile_origin(tn, io_n, oo_n, lv_n);
#pragma unroll
      for (int s = 0; s < KH; ++s) bn[s] = (2 * s < K) ? in[io_n + (unsigned)(2 * s + half) * (unsigned)P] : 0.f;
    }
    for (int nt = 0; nt < nblk; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* __restrict__ wr = Ws + nt * 32 * WS + a_lane;
#pragma unroll
      for (int s = 0; s < KH; ++s)
        if (2 * s < K) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[2 * s], bc[s], acc, 0, 0, 0);
      epilogue(acc, nt, oo, lv);
    }
#pragma unroll
    for (int s = 0; s < KH; ++s) bc[s] = bn[s];
    io = io_n; oo = oo_n; lv = lv_n;
  }
}

// Host side.  Returns false when the layer stays with the tiled kernel (conv_mfma<1,...>).
static int g_pw_mode = 0;        // test / tuning hook: 0 = by the measured rule, 1 = never, 2 = whenever the shape is covered
void pw_set_mode(int m) { g_pw_mode = m; }

bool pw_launch(const float* in, const float* w, const float* bias, const float* res, float* out, int B, int K, int N, int P,
               int act, bool dgrad, hipStream_t s) {
  if (g_pw_mode == 1) return false;
  if (K % 2 != 0 || K < 8 || K > 64 || N < 8) return false;
  if ((long)B * P * (K > N ? K : N) >= (1L << 31)) return false;              // 32-bit element offsets
  if (g_pw_mode == 0 && ((long)B * P < 32768 || (dgrad && K > 32))) return false;
  const long ntile = ((long)B * P + 31) / 32;
  const unsigned gx = (unsigned)std::min<long>((ntile + 3) / 4, 512);
  const int n_per_wg = (N + 31) / 32 * 32;
  const size_t lds = sizeof(float) * (size_t)n_per_wg * (K + 2);
  if (lds > 64 * 1024) return false;                                            // two workgroups per CU
  const dim3 grid(gx, 1);
  if (K <= 32) hipLaunchKernelGGL((pw_mfma<16>), grid, dim3(256), lds, s, in, w, bias, res, out, B, K, N, P, act, dgrad ? 1 : 0, n_per_wg);
  else         hipLaunchKernelGGL((pw_mfma<32>), grid, dim3(256), lds, s, in, w, bias, res, out, B, K, N, P, act, dgrad ? 1 : 0, n_per_wg);
  return true;
}

}  // namespace afd

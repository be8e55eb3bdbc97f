// wino.hip -- F5: 3x3 (pad 1) convolution forward / dgrad by Winograd F(2x2, 3x3) on the exact-fp32
// matrix cores.  Y = A^T [ sum_k (G g G^T) .* (B^T d B) ] A : every 2x2 output tile costs 16 multiplies per
// (cin, cout) pair instead of 36, so the MFMA work of a 3x3 layer drops 2.25x; everything stays fp32
// (inputs, products, accumulation), only the association of the sums differs from the direct form.
//
//   wino_weights   U[k/8][xi][n][8] = (G g G^T)[xi]   (xi = 4i + j; the 8 channels of a chunk stored in MFMA
//                  fragment order), once per call; for dgrad the taps are flipped and the channel roles swapped,
//                  so ONE main kernel serves both passes.
//   conv_wino      a workgroup owns BN output channels x 64 tiles (= 256 output pixels).  Per chunk of 8 input
//                  channels: the haloed input rows and the U chunk are staged global -> registers -> LDS (the
//                  next chunk's loads are in flight during the multiplies); each lane transforms the two
//                  (tile, channel) 4x4 patches that ARE its B operands (2 x 32 adds, V never exists in memory),
//                  then the wave multiplies its 32 channels x 16 tiles for all 16 xi with
//                  v_mfma_f32_16x16x4_f32 (128 accumulator registers).
//                  The 16 xi of one (channel, tile) sit in ONE lane, so the output transform A^T M A is
//                  register arithmetic and the 2x2 results leave as float2 stores.
#include <cstdlib>
#include "common.h"
#include "bf3_weights.h"
#include "h2_common.h"

namespace afd {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// geometry of an NT-tile workgroup on a square S x S map: either TROWS rows of tiles of ONE image (NT <= tiles per
// image) or TI whole images.  The haloed input rows of the group live in LDS as [channel][TI][RROWS][Wp].
template <int S, int NT> struct WGeo {
  static constexpr int W = S, TPR = S / 2, TPI = TPR * TPR;                 // tiles per row / per image
  static constexpr int TI = NT <= TPI ? 1 : NT / TPI;                       // images per group
  static constexpr int TROWS = NT <= TPI ? NT / TPR : TPR;                  // tile rows per image in the group
  static constexpr int GPI = NT <= TPI ? TPI / NT : 1;                      // groups per image
  static constexpr int PXR = 2 * TROWS, Wp = S + 2, IMG = (PXR + 2) * Wp, RS = TI * IMG;
  static constexpr int RSP = RS + ((32 - RS % 64) + 64) % 64;               // channel stride = 32 mod 64 banks: the two channels of a b64 lane group never collide
  static_assert(NT % TPR == 0 && (NT <= TPI ? TPI % NT == 0 : NT % TPI == 0), "tile group must be whole tile rows or whole images");
};

// hand-issued LDS fragment reads: two ds_read_b64 at base + immediate offsets, and the counted wait that makes their
// results usable (the registers go through the wait statement, so no consumer can be scheduled above it)
template <int OFF0, int OFF1>
__device__ __forceinline__ void lds_issue_pair(f32x2& a, f32x2& b, unsigned base) {
  asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=&v"(a), "=&v"(b) : "v"(base), "n"(OFF0), "n"(OFF1));
}
template <int N>
__device__ __forceinline__ void lds_wait(f32x2& a, f32x2& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

__device__ __forceinline__ int kperm8(int k) { return 2 * (k & 3) + (k >> 2); }   // LDS slot of channel k of a chunk: lane quarter q reads k = q, q + 4 as one b64

// Transformed weights of one 3x3 layer, either or both passes in one launch:
//   Uf[(ci>>3)][xi][co][kperm8(ci&7)] = (G g G^T)[xi],  g = w[co][ci]                      (forward: reduction over ci)
//   Ud[(co>>3)][xi'][ci][kperm8(co&7)] = (G g' G^T)[xi'], g' = w[co][ci] rotated by 180 deg  (dgrad: reduction over co)
// G J = P G with P swapping rows 0 and 3 (J the 3x3 flip), so the dgrad form is the forward form with the xi indices
// permuted: no second transform.
__device__ __forceinline__ void wino_weights_block(const float* __restrict__ w, float* __restrict__ Uf, float* __restrict__ Ud,
                                                   int Cin, int Cout, int blk, int lane) {
  // a wave = one 8 x 8 block (8 output x 8 input channels): in BOTH images the 64 values of a transform index form one
  // contiguous 256-byte run, so every store is coalesced
  const int nci8 = Cin >> 3;
  if (blk >= nci8 * (Cout >> 3)) return;
  const int ci = (blk % nci8) * 8 + (lane & 7), co = (blk / nci8) * 8 + (lane >> 3);
  const float* p = w + ((long)co * Cin + ci) * 9;
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = p[t];
  // t = G g  (4x3), u = t G^T (4x4);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
  float t[4][3], u[16];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
    t[0][c] = g0;
    t[1][c] = 0.5f * (g0 + g1 + g2);
    t[2][c] = 0.5f * (g0 - g1 + g2);
    t[3][c] = g2;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = t[i][0], b = t[i][1], c = t[i][2];
    u[4 * i + 0] = a;
    u[4 * i + 1] = 0.5f * (a + b + c);
    u[4 * i + 2] = 0.5f * (a - b + c);
    u[4 * i + 3] = c;
  }
  if (Uf) {
    float* o = Uf + ((long)(ci >> 3) * 16 * Cout + co) * 8 + kperm8(ci & 7);
    const long xs = (long)Cout * 8;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) o[xi * xs] = u[xi];
  }
  if (Ud) {
    float* o = Ud + ((long)(co >> 3) * 16 * Cin + ci) * 8 + kperm8(co & 7);
    const long xs = (long)Cin * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pi = i == 0 ? 3 : (i == 3 ? 0 : i), pj = j == 0 ? 3 : (j == 3 ? 0 : j);
        o[(4 * i + j) * xs] = u[4 * pi + pj];
      }
  }
}

__global__ __launch_bounds__(256) void wino_weights(const float* __restrict__ w, float* __restrict__ Uf, float* __restrict__ Ud,
                                                    int Cin, int Cout) {
  wino_weights_block(w, Uf, Ud, Cin, Cout, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// every layer of a model in ONE launch: workgroup g works on layer wg_desc[g] (afd_wino_desc, include/afd.h)

__global__ __launch_bounds__(256) void wino_weights_batched(const afd_wino_desc* __restrict__ descs, const int* __restrict__ wg_desc) {
  const afd_wino_desc d = descs[wg_desc[blockIdx.x]];
  const int blk = (blockIdx.x - d.first_wg) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* wf = (d.kinds & 1) ? nullptr : d.u_fwd; float* wd = (d.kinds & 2) ? nullptr : d.u_dgrad;
  if (wf || wd) wino_weights_block(d.w, wf, wd, d.Cin, d.Cout, blk, lane);
  void* bf = (d.kinds & 1) ? d.u_fwd : nullptr; void* bd = (d.kinds & 2) ? d.u_dgrad : nullptr;
  if (bf || bd) {                                     // the direct form's image: f16x2 (h2.hip) or, kinds bit 2, bf16x3 (bf3.hip); same wave-per-8x8-block decomposition
    if (d.kinds & 4) bf3_weights_block(d.w, static_cast<__bf16*>(bf), static_cast<__bf16*>(bd), d.Cin, d.Cout, blk, lane);
    else h2_weights_block(d.w, bf, bd, d.Cin, d.Cout, blk, lane);
  }
}
// the per-row scales of the f16x2 images, one launch ahead of the one above over the same grid: a layer's Cin*Cout/64
// waves share its Cin + Cout rows (one wave per row on the wide layers)
__global__ __launch_bounds__(256) void h2_wscale_batched(const afd_wino_desc* __restrict__ descs, const int* __restrict__ wg_desc) {
  const afd_wino_desc d = descs[wg_desc[blockIdx.x]];
  if (d.kinds & 4) return;
  void* bf = (d.kinds & 1) ? d.u_fwd : nullptr; void* bd = (d.kinds & 2) ? d.u_dgrad : nullptr;
  const int nw = 4 * ((d.Cin * d.Cout / 64 + 3) / 4);                  // the waves this layer owns in the grid (afd_wino_desc: ceil(Cin*Cout/256) workgroups)
  if (bf || bd) h2_wscale_rows(d.w, bf, bd, d.Cin, d.Cout, (blockIdx.x - d.first_wg) * 4 + (threadIdx.x >> 6), nw, threadIdx.x & 63);
}

template <int GEO, int BN, int NT>
__global__ __launch_bounds__(BN * NT / 8, BN * NT == 4096 ? 1 : 2) void conv_wino(const float* __restrict__ x, const float* __restrict__ U,
                                                    const float* __restrict__ bias, const float* __restrict__ res,
                                                    float* __restrict__ y, int B, int K, int N, int act, int items) {
  using G = WGeo<GEO, NT>;
  constexpr int KC = 8, SU = 10, NTH = BN * NT / 8;                        // one wave per 32 channels x 16 tiles
  constexpr int W = G::W, H = G::W, HW = W * W, Wp = G::Wp, IMG = G::IMG, RS = G::RS, RSP = G::RSP;
  constexpr int NPOS = (RS + NTH - 1) / NTH;         // raw positions per thread
  constexpr int NUV = 16 * BN * 2 / NTH;             // float4 pieces of the U chunk per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = 16 * BN * SU + KC * RSP;       // floats per stage: U image [row block][16 xi][16 rows][SU], then the
  float* Us = smem;                                  // haloed input rows [KC][RSP]; TWO stages: chunk c+1 is written while
  float* Rs = Us + 16 * BN * SU;                     // chunk c is multiplied, one barrier per chunk
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int NB = N / BN;                             // work item w = (tile group, channel block): w % NB is the channel block

  // ---- staging plan of the item being FETCHED (it runs ahead of the item being multiplied across item seams)
  unsigned rsrc[NPOS]; unsigned rvalid = 0; unsigned usrc0 = 0;
  constexpr int XSTEP = NTH / (2 * BN);
  static_assert(NTH % (2 * BN) == 0 && XSTEP * NUV == 16, "U chunk must split into whole xi planes per pass");
  const int u_h = tid & 1, u_n = (tid >> 1) % BN, u_xi = tid / (2 * BN);
  const int udst0 = (((u_n >> 4) * 16 + u_xi) * 16 + (u_n & 15)) * SU + 4 * u_h;   // LDS image [row block][xi][16 rows][SU]
  const long ustep = (long)XSTEP * N * 8;
  auto plan = [&](int w) {
    const int tg = w / NB, n0 = (w % NB) * BN;
    const int b0 = (G::TI == 1) ? tg / G::GPI : tg * G::TI;
    const int row0 = (G::TI == 1) ? (tg % G::GPI) * G::PXR : 0;
    rvalid = 0;
#pragma unroll
    for (int e = 0; e < NPOS; ++e) {
      const int pos = tid + NTH * e;
      bool ok = false; unsigned src = 0;
      if (pos < RS) {
        const int ti = pos / IMG, rem = pos % IMG;
        const int yy = row0 + rem / Wp - 1, xx = rem % Wp - 1, b = b0 + ti;
        ok = b < B && yy >= 0 && yy < H && xx >= 0 && xx < W;
        if (ok) src = (unsigned)(b * K) * (unsigned)HW + (unsigned)(yy * W + xx);
      }
      rsrc[e] = src; rvalid |= (ok ? 1u : 0u) << e;
    }
    usrc0 = (unsigned)((u_xi * N + n0 + u_n) * 8 + 4 * u_h);
  };
  // ---- wave roles: 32 channels x 16 tiles, all 16 xi.  Lane (li, lq): A rows wn*32 + {li, 16 + li}, tile wt*16 + li,
  // channels lq and lq + 4 of the chunk (the two k-steps of v_mfma_f32_16x16x4_f32)
  const int wn = wv % (BN / 32), wt = wv / (BN / 32);
  const int li = lane & 15, lq = lane >> 4;
  const int tt = wt * 16 + li;
  const int poff = (tt / (G::TROWS * G::TPR)) * IMG + 2 * ((tt / G::TPR) % G::TROWS) * Wp + 2 * (tt % G::TPR);

  // two register sets: the loads of chunk c+2 are issued right after chunk c+1 left its set for LDS, i.e. every load
  // has TWO chunk periods to land (one period is about the loaded-chip memory latency: with a single set the waves
  // parked on s_waitcnt for a fifth of their time)
  struct Regs { float r[NPOS][KC]; f32x4 u[NUV]; unsigned ok; };
  Regs S0, S1;
  auto fetch = [&](Regs& S, int c) {
    const float* __restrict__ xk = x + (long)c * KC * HW;
    const float* __restrict__ uk = U + (long)c * 16 * N * 8;
#pragma unroll
    for (int e = 0; e < NUV; ++e) S.u[e] = *reinterpret_cast<const f32x4*>(uk + e * ustep + usrc0);
#pragma unroll
    for (int e = 0; e < NPOS; ++e)
      if (tid + NTH * e < RS) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) S.r[e][kc] = (xk + kc * HW)[rsrc[e]];      // scalar base per channel, one lane offset
      }
    S.ok = rvalid;
  };
  auto commit = [&](const Regs& S, int stage) {
    float* Us = smem + stage * BUF;
    float* Rs = Us + 16 * BN * SU;
#pragma unroll
    for (int e = 0; e < NUV; ++e) {
      float2* d = reinterpret_cast<float2*>(Us + udst0 + e * (XSTEP * 16 * SU));
      d[0] = make_float2(S.u[e][0], S.u[e][1]);
      d[1] = make_float2(S.u[e][2], S.u[e][3]);
    }
#pragma unroll
    for (int e = 0; e < NPOS; ++e)
      if (tid + NTH * e < RS) {
        const bool ok = (S.ok >> e) & 1u;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) Rs[kc * RSP + tid + NTH * e] = ok ? S.r[e][kc] : 0.f;
      }
  };
  // V = B^T d B for one 4x4 patch; B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
  auto patch = [&](const float* Rs, int k, float (&v)[16]) {
    const float* r = Rs + k * RSP + poff;
    float d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float2 lo = *reinterpret_cast<const float2*>(r + a * Wp);
      const float2 hi = *reinterpret_cast<const float2*>(r + a * Wp + 2);
      d[a][0] = lo.x; d[a][1] = lo.y; d[a][2] = hi.x; d[a][3] = hi.y;
    }
    float w[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[0][j] = d[0][j] - d[2][j];
      w[1][j] = d[1][j] + d[2][j];
      w[2][j] = d[2][j] - d[1][j];
      w[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[4 * i + 0] = w[i][0] - w[i][2];
      v[4 * i + 1] = w[i][1] + w[i][2];
      v[4 * i + 2] = w[i][2] - w[i][1];
      v[4 * i + 3] = w[i][1] - w[i][3];
    }
  };
  // byte address (LDS offset) of the lane's A fragment: rows li of the wave's two 16-row blocks, channels lq and lq + 4.
  // The fragment reads are hand-issued ds_read_b64 with immediate offsets (xi * 640 B, second block + 10240 B): the
  // compiler's merged ds_read2 forms need a fresh base register per xi.
  const unsigned ap = (unsigned)(size_t)(Us + (wn * 2 * 256 + li) * SU + 2 * lq);
  const int o_ti = tt / (G::TROWS * G::TPR), o_ty = (tt / G::TPR) % G::TROWS, o_tx = tt % G::TPR;
  const int nchunks = K / KC;

  // ---- loop over work items (one per workgroup by default; the grid may be smaller: tools)
  for (int w = blockIdx.x; w < items; w += gridDim.x) {
    plan(w);
    fetch(S0, 0);
    if (nchunks > 1) fetch(S1, 1);
    f32x4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[xi][h] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto multiply = [&](int st) {
      const float* Rc = Rs + st * BUF;
      const unsigned apc = ap + (unsigned)(st * BUF * 4);
      // A fragments run one xi ahead of the multiplies (the first pair is in flight under the transform)
      f32x2 alo[2], ahi[2];
      lds_issue_pair<0, 256 * SU * 4>(alo[0], ahi[0], apc);
      // the wave's own B operands: V = B^T d B of its 16 tiles x 8 channels, straight into registers (the waves that
      // share these tiles repeat the 32 adds; no V image in LDS)
      float v0[16], v1[16];
      patch(Rc, lq, v0);
      patch(Rc, lq + 4, v1);
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        constexpr int STEP = 16 * SU * 4;
        f32x2& lo = alo[xi & 1]; f32x2& hi = ahi[xi & 1];
        if (xi < 15) {
          switch (xi) {                                   // (immediates must be literal per instruction)
#define AFD_NEXT(X) case X: lds_issue_pair<(X + 1) * STEP, (X + 1) * STEP + 256 * SU * 4>(alo[(X + 1) & 1], ahi[(X + 1) & 1], apc); break;
            AFD_NEXT(0) AFD_NEXT(1) AFD_NEXT(2) AFD_NEXT(3) AFD_NEXT(4) AFD_NEXT(5) AFD_NEXT(6) AFD_NEXT(7)
            AFD_NEXT(8) AFD_NEXT(9) AFD_NEXT(10) AFD_NEXT(11) AFD_NEXT(12) AFD_NEXT(13) AFD_NEXT(14)
#undef AFD_NEXT
          }
          lds_wait<2>(lo, hi);                            // this xi's pair landed; the next pair stays in flight
        } else {
          lds_wait<0>(lo, hi);
        }
        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(lo[0], v0[xi], acc[xi][0], 0, 0, 0);
        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(hi[0], v0[xi], acc[xi][1], 0, 0, 0);
        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(lo[1], v1[xi], acc[xi][0], 0, 0, 0);
        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(hi[1], v1[xi], acc[xi][1], 0, 0, 0);
      }
    };

    // chunk c lives in register set c & 1 and in LDS stage c & 1
    __syncthreads();                                    // (a previous item's last multiplies are done with stage 0)
    commit(S0, 0);
    if (nchunks > 2) fetch(S0, 2);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
      multiply(0);
      if (c + 1 < nchunks) {                            // chunk c+1 -> stage 1 while slower waves still multiply chunk c
        commit(S1, 1);
        if (c + 3 < nchunks) fetch(S1, c + 3);
        __syncthreads();
        multiply(1);
        if (c + 2 < nchunks) {
          commit(S0, 0);
          if (c + 4 < nchunks) fetch(S0, c + 4);
          __syncthreads();
        }
      }
    }

    // ---- output transform Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]) and the epilogue.  Lane: tile wt*16 + li,
    // channels n0 + wn*32 + 16*h + 4*lq + r.
    const int tg = w / NB, n0 = (w % NB) * BN;
    const int ob = ((G::TI == 1) ? tg / G::GPI : tg * G::TI) + o_ti;
    const int row0 = (G::TI == 1) ? (tg % G::GPI) * G::PXR : 0;
    if (ob < B) {
      const int opix = (row0 + 2 * o_ty) * W + 2 * o_tx;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wn * 32 + 16 * h + 4 * lq + r;
          float tcol[4][2];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float m0 = acc[4 * i + 0][h][r], m1 = acc[4 * i + 1][h][r], m2 = acc[4 * i + 2][h][r], m3 = acc[4 * i + 3][h][r];
            tcol[i][0] = (m0 + m1) + m2;
            tcol[i][1] = (m1 - m2) - m3;
          }
          float y00 = (tcol[0][0] + tcol[1][0]) + tcol[2][0], y01 = (tcol[0][1] + tcol[1][1]) + tcol[2][1];
          float y10 = (tcol[1][0] - tcol[2][0]) - tcol[3][0], y11 = (tcol[1][1] - tcol[2][1]) - tcol[3][1];
          const unsigned idx = (unsigned)((ob * N + n) * HW + opix);          // < 2^31 (host check)
          if (bias) { const float bb = bias[n]; y00 += bb; y01 += bb; y10 += bb; y11 += bb; }
          if (act == 1) { y00 = gelu_erf(y00); y01 = gelu_erf(y01); y10 = gelu_erf(y10); y11 = gelu_erf(y11); }
          if (res) {
            const float2 r0 = *reinterpret_cast<const float2*>(res + idx), r1 = *reinterpret_cast<const float2*>(res + idx + W);
            y00 += r0.x; y01 += r0.y; y10 += r1.x; y11 += r1.y;
          }
          *reinterpret_cast<float2*>(y + idx) = make_float2(y00, y01);
          *reinterpret_cast<float2*>(y + idx + W) = make_float2(y10, y11);
        }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Small maps (4x4, 8x8): few tiles, deep reductions.  A workgroup owns 32 output channels x 16 tiles (whole images, so
// every halo cell is a constant zero) and its FOUR WAVES SPLIT THE INPUT CHANNELS: each wave runs the whole
// F(2x2,3x3) pipeline on its quarter of K with no workgroup barrier in the loop -- its 8 input channels of a chunk sit
// in a wave-private LDS patch image, and its A fragments come STRAIGHT FROM GLOBAL MEMORY: in the [k/8][xi][n][8]
// weight image the 64 lanes of a fragment read one contiguous 512-byte run.  The four partial results meet once, after
// the (linear) output transform, through LDS.
// ------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256, 2) void conv_wino_sk(const float* __restrict__ x, const float* __restrict__ U,
                                                       const float* __restrict__ bias, const float* __restrict__ res,
                                                       float* __restrict__ y, int B, int K, int N, int act) {
  using G = WGeo<S, 16>;
  static_assert(G::GPI == 1, "tile groups must be whole images");
  constexpr int KC = 8, W = S, HW = S * S, Wp = G::Wp, IMG = G::IMG, RS = G::RS, RSP = G::RSP, TI = G::TI;
  constexpr int UPC = TI * S * (S / 4);              // float4 pieces per channel (= 16)
  static_assert(UPC == 16, "one chunk = 128 float4 pieces per wave");
  constexpr int DEPTH = 8;                           // A fragments in flight, in units of xi
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* Rs = smem + wv * (KC * RSP);                // this wave's patch image [KC][RSP]
  float* red = smem + 4 * KC * RSP;                  // [4 waves][32 values][64 lanes]
  const int n0 = blockIdx.x * 32, tg = blockIdx.y, b0 = tg * TI;
  const int li = lane & 15, lq = lane >> 4;
  const int poff = (li / (G::TROWS * G::TPR)) * IMG + 2 * ((li / G::TPR) % G::TROWS) * Wp + 2 * (li % G::TPR);
  const int KS = K / 4, nchunks = KS / KC, kw = wv * KS;       // this wave's channel slice

  for (int i = lane; i < KC * RSP; i += 64) Rs[i] = 0.f;      // halo cells stay zero for the whole kernel

  // staging: piece u = lane + 64 e (e = 0, 1): channel u / 16, float4 (u % 16) of the TI x S x S block
  unsigned rsrc[2]; int rdst[2]; bool rok[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int u = lane + 64 * e, ch = u / UPC, q = u % UPC;
    const int ti = q / (S * (S / 4)), rr = (q / (S / 4)) % S, x4 = q % (S / 4);
    rok[e] = b0 + ti < B;
    rsrc[e] = rok[e] ? (unsigned)(((b0 + ti) * K + kw + ch) * HW + rr * W + 4 * x4) : 0u;
    rdst[e] = ch * RSP + ti * IMG + (rr + 1) * Wp + 1 + 4 * x4;
  }
  f32x4 rreg[2];
  auto fetch = [&](int c) {
    const float* xk = x + (long)c * KC * HW;
#pragma unroll
    for (int e = 0; e < 2; ++e) rreg[e] = *reinterpret_cast<const f32x4*>(xk + rsrc[e]);
  };
  auto commit = [&]() {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float* d = Rs + rdst[e];
      const bool ok = rok[e];
      d[0] = ok ? rreg[e][0] : 0.f; d[1] = ok ? rreg[e][1] : 0.f; d[2] = ok ? rreg[e][2] : 0.f; d[3] = ok ? rreg[e][3] : 0.f;
    }
  };
  auto patch = [&](int k, float (&v)[16]) {
    const float* r = Rs + k * RSP + poff;
    float d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float2 lo = *reinterpret_cast<const float2*>(r + a * Wp);
      const float2 hi = *reinterpret_cast<const float2*>(r + a * Wp + 2);
      d[a][0] = lo.x; d[a][1] = lo.y; d[a][2] = hi.x; d[a][3] = hi.y;
    }
    float w[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[0][j] = d[0][j] - d[2][j];
      w[1][j] = d[1][j] + d[2][j];
      w[2][j] = d[2][j] - d[1][j];
      w[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[4 * i + 0] = w[i][0] - w[i][2];
      v[4 * i + 1] = w[i][1] + w[i][2];
      v[4 * i + 2] = w[i][2] - w[i][1];
      v[4 * i + 3] = w[i][1] - w[i][3];
    }
  };
  // A fragment of (chunk c, xi): U[(kw/8 + c)][xi][n0 + li (+16)][2 lq .. 2 lq + 1]; the lanes of one load cover 512 B
  const float* ua = U + ((long)(kw / KC) * 16 * N + n0 + li) * 8 + 2 * lq;
  const long xstep = (long)N * 8;                    // one xi plane
  const int total = nchunks * 16;                    // flat (chunk, xi) index g = 16 c + xi
  f32x2 abuf[DEPTH][2];
  auto aload = [&](int g, int slot) {
    const float* p = ua + (long)g * xstep;           // (chunk c, xi) planes are consecutive: (16 c + xi) * N * 8
    abuf[slot][0] = *reinterpret_cast<const f32x2*>(p);
    abuf[slot][1] = *reinterpret_cast<const f32x2*>(p + 128);
  };

  f32x4 acc[16][2];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[xi][h] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int g = 0; g < DEPTH; ++g) aload(g, g);       // (total >= 16 > DEPTH)
  fetch(0);
  for (int c = 0; c < nchunks; ++c) {
    commit();                                        // wave-private image: program order is the only ordering needed
    if (c + 1 < nchunks) fetch(c + 1);
    float v0[16], v1[16];
    patch(lq, v0);
    patch(lq + 4, v1);
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      const f32x2 lo = abuf[xi % DEPTH][0], hi = abuf[xi % DEPTH][1];
      const int gn = c * 16 + xi + DEPTH;
      if (gn < total) aload(gn, xi % DEPTH);
      __builtin_amdgcn_sched_barrier(0);             // keep the refill above the multiplies
      acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(lo[0], v0[xi], acc[xi][0], 0, 0, 0);
      acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(hi[0], v0[xi], acc[xi][1], 0, 0, 0);
      acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(lo[1], v1[xi], acc[xi][0], 0, 0, 0);
      acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(hi[1], v1[xi], acc[xi][1], 0, 0, 0);
    }
  }

  // ---- output transform of the wave's partial sums (linear, so it commutes with the cross-wave sum), then the four
  // partials meet in LDS: value index j = (h*4 + r)*4 + {y00, y01, y10, y11}
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float tcol[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float m0 = acc[4 * i + 0][h][r], m1 = acc[4 * i + 1][h][r], m2 = acc[4 * i + 2][h][r], m3 = acc[4 * i + 3][h][r];
        tcol[i][0] = (m0 + m1) + m2;
        tcol[i][1] = (m1 - m2) - m3;
      }
      float* o = red + ((wv * 32 + (h * 4 + r) * 4) * 64) + lane;
      o[0] = (tcol[0][0] + tcol[1][0]) + tcol[2][0];
      o[64] = (tcol[0][1] + tcol[1][1]) + tcol[2][1];
      o[128] = (tcol[1][0] - tcol[2][0]) - tcol[3][0];
      o[192] = (tcol[1][1] - tcol[2][1]) - tcol[3][1];
    }
  __syncthreads();
  // wave w finishes the (h, r) pairs 2w and 2w + 1 (fixed summation order: deterministic)
  const int ob = b0 + li / (G::TROWS * G::TPR);
  if (ob >= B) return;
  const int opix = 2 * ((li / G::TPR) % G::TROWS) * W + 2 * (li % G::TPR);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int hr = 2 * wv + e, h = hr >> 2, r = hr & 3;
    float yv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* q = red + (hr * 4 + j) * 64 + lane;
      yv[j] = (q[0] + q[32 * 64]) + (q[2 * 32 * 64] + q[3 * 32 * 64]);
    }
    const int n = n0 + 16 * h + 4 * lq + r;
    const unsigned idx = (unsigned)((ob * N + n) * HW + opix);
    if (bias) { const float bb = bias[n]; yv[0] += bb; yv[1] += bb; yv[2] += bb; yv[3] += bb; }
    if (act == 1) { yv[0] = gelu_erf(yv[0]); yv[1] = gelu_erf(yv[1]); yv[2] = gelu_erf(yv[2]); yv[3] = gelu_erf(yv[3]); }
    if (res) {
      const float2 r0 = *reinterpret_cast<const float2*>(res + idx), r1 = *reinterpret_cast<const float2*>(res + idx + W);
      yv[0] += r0.x; yv[1] += r0.y; yv[2] += r1.x; yv[3] += r1.y;
    }
    *reinterpret_cast<float2*>(y + idx) = make_float2(yv[0], yv[1]);
    *reinterpret_cast<float2*>(y + idx + W) = make_float2(yv[2], yv[3]);
  }
}

template <int S>
static void wino_sk_launch(const float* x, const float* U, const float* bias, const float* res, float* y, int B, int K, int N, int act,
                           hipStream_t s) {
  using G = WGeo<S, 16>;
  const size_t lds = sizeof(float) * (4 * 8 * G::RSP + 4 * 32 * 64);
  const int tg = (B + G::TI - 1) / G::TI;
  hipLaunchKernelGGL((conv_wino_sk<S>), dim3(N / 32, tg), dim3(256), lds, s, x, U, bias, res, y, B, K, N, act);
}

static int g_wino_grid = 0;      // test / tuning hook: persistent grid size (0 = fill the chip once)
void wino_set_grid(int g) { g_wino_grid = g; }
static int g_wino_mode = 0;        // 0 = by rule, 1 = off, 2..5 = whenever the shape is supported: workgroups of 64x64, 32x64, 64x32, 32x32 (channels x tiles); 6 = the small-map split-K kernel wherever covered
void wino_set_mode(int m) { g_wino_mode = m; }

// workgroup shape for a Winograd launch: BN * 256 + NT, or 0 when the layer stays on the direct kernel
int wino_plan(int B, int K, int N, int H, int W) {
  if (g_wino_mode == 1) return 0;
  if (H != W || (W != 64 && W != 32 && W != 16 && W != 8 && W != 4)) return 0;
  if (K % 8 || N % 32 || K < 8) return 0;
  if ((long)B * K * H * W >= (1L << 31) || (long)B * N * H * W >= (1L << 31) || (long)16 * N * K >= (1L << 28)) return 0;
  const bool n64 = N % 64 == 0;
  if (g_wino_mode == 2) return (n64 ? 64 : 32) * 256 + 64;
  if (g_wino_mode == 3) return 32 * 256 + 64;
  if (g_wino_mode == 4) return (n64 ? 64 : 32) * 256 + 32;
  if (g_wino_mode == 5) return 32 * 256 + 32;
  // measured per layer at B = 256 (tools/wino_bench.py): 64 channels x 64 tiles (8 waves, one workgroup per CU, the input
  // transform shared by two channel blocks) is best once it gives every CU a workgroup; below that 32 x 64 (two
  // workgroups per CU); Winograd wins from 4 chunks up; the 4x4 maps (4 tiles per image) and thin launches stay direct
  const long tiles = (long)B * (H / 2) * (W / 2);
  if (g_wino_mode == 6) return (W <= 8 && K % 32 == 0 && K >= 64) ? 1 : 0;       // (tests) the small-map kernel wherever covered
  // 4x4 and 8x8 maps whose launch cannot give every CU a 64-tile workgroup: the in-workgroup split-K kernel
  if (W <= 8 && K % 32 == 0 && K >= 64 && !(n64 && (tiles / 64) * (N / 64) >= 256) && (tiles / 16) * (N / 32) >= 128) return 1;
  if (K < 32 || W < 8) return 0;
  if (n64 && (tiles / 64) * (N / 64) >= 256) return 64 * 256 + 64;
  if ((tiles / 64) * (N / 32) >= 256) return 32 * 256 + 64;
  return 0;
}

template <int GEO, int BN, int NT>
static void wino_launch_t(const float* x, const float* U, const float* bias, const float* res, float* y, int B, int K, int N, int act,
                          hipStream_t s) {
  using G = WGeo<GEO, NT>;
  const size_t lds = 2 * sizeof(float) * (16 * BN * 10 + 8 * G::RSP);   // two stages
  (void)lds_opt_in(&conv_wino<GEO, BN, NT>, lds);                    // (> 64 KB of dynamic LDS: once per device and kernel)
  const int tg = G::TI == 1 ? B * G::GPI : (B + G::TI - 1) / G::TI;
  const int items = tg * (N / BN);
  // persistent grid: as many workgroups as fit the chip at once (8-wave groups: one per CU, smaller ones: two), each
  // walking items g, g + grid, ...  (g_wino_grid overrides, tools/wino_bench.py)
  // one item per workgroup by default: with two workgroups per CU the hardware's own dispatch already overlaps one
  // group's prologue / store tail with the other's multiplies, and balances better than a fixed walk (measured);
  // g_wino_grid > 0 makes the groups persistent (tools)
  int grid = g_wino_grid > 0 ? g_wino_grid : items;
  if (grid > items) grid = items;
  hipLaunchKernelGGL((conv_wino<GEO, BN, NT>), dim3(grid), dim3(BN * NT / 8), lds, s, x, U, bias, res, y, B, K, N, act, items);
}

// the direct forms (take the layers they cover ahead of the Winograd kernels): f16x2 (h2.hip, round 3) or bf16x3 (bf3.hip)
bool bf3_ok(int B, int K, int N, int H, int W);
bool direct_form_is_bf3();
void bf3_weights_launch(const float* w, void* Wf, void* Wd, int Cin, int Cout, hipStream_t s);
void bf3_conv(const float* x, const void* Wp, const float* bias, const float* res, float* y, int B, int K, int N, int S, int act,
              hipStream_t s);
void h2_weights_launch(const float* w, void* Wf, void* Wd, int Cin, int Cout, hipStream_t s);
void h2_conv(const float* x, const void* Wp, const float* bias, const float* res, float* y, int B, int K, int N, int S, int act,
             hipStream_t s, bool small);
int direct_plan(int B, int K, int N, int H, int W);

// conv (dgrad = false: x (B,K,H,W), w (N,K,3,3)) or its input gradient (dgrad = true: x = dY (B,K=Cout,H,W), w (K,N,3,3))
bool wino_conv(const float* x, const float* w, const float* bias, const float* res, float* y, float* U, int B, int K, int N, int H,
               int W, int act, bool dgrad, bool weights_ready, hipStream_t s) {
  const int Cin = dgrad ? N : K, Cout = dgrad ? K : N;
  if (U && bf3_ok(B, K, N, H, W)) {                 // the workspace then holds the direct form's weight image of this pass
    if (direct_form_is_bf3()) {
      if (!weights_ready) bf3_weights_launch(w, dgrad ? nullptr : U, dgrad ? U : nullptr, Cin, Cout, s);
      bf3_conv(x, U, bias, res, y, B, K, N, W, act, s);
    } else {
      if (!weights_ready) h2_weights_launch(w, dgrad ? nullptr : U, dgrad ? U : nullptr, Cin, Cout, s);
      h2_conv(x, U, bias, res, y, B, K, N, W, act, s, direct_plan(B, K, N, H, W) == 2);
    }
    return true;
  }
  const int plan = wino_plan(B, K, N, H, W);
  if (!plan || !U) return false;
  const int bn = plan >> 8, nt = plan & 255;
  if (plan == 1) {
    if (!weights_ready)
      hipLaunchKernelGGL(wino_weights, dim3((unsigned)((K * N / 64 + 3) / 4)), dim3(256), 0, s, w, dgrad ? nullptr : U, dgrad ? U : nullptr, Cin, Cout);
    if (W == 4) wino_sk_launch<4>(x, U, bias, res, y, B, K, N, act, s); else wino_sk_launch<8>(x, U, bias, res, y, B, K, N, act, s);
    return true;
  }
  if (!weights_ready)
    hipLaunchKernelGGL(wino_weights, dim3((unsigned)((K * N / 64 + 3) / 4)), dim3(256), 0, s, w, dgrad ? nullptr : U, dgrad ? U : nullptr, Cin, Cout);
#define AFD_WINO(GEO_)                                                                            \
  if (bn == 64 && nt == 64) wino_launch_t<GEO_, 64, 64>(x, U, bias, res, y, B, K, N, act, s);     \
  else if (bn == 32 && nt == 64) wino_launch_t<GEO_, 32, 64>(x, U, bias, res, y, B, K, N, act, s); \
  else if (bn == 64) wino_launch_t<GEO_, 64, 32>(x, U, bias, res, y, B, K, N, act, s);            \
  else wino_launch_t<GEO_, 32, 32>(x, U, bias, res, y, B, K, N, act, s)
  if (W == 64) { AFD_WINO(64); } else if (W == 32) { AFD_WINO(32); } else if (W == 16) { AFD_WINO(16); } else if (W == 8) { AFD_WINO(8); } else { AFD_WINO(4); }
#undef AFD_WINO
  return true;
}


// ------------------------------------------------------------------------------------------
// wgrad by the adjoint of the same factorisation:  dU[xi][n][k] = sum_tiles (A dY A^T)[xi][n][tile] * (B^T d B)[xi][k][tile],
// dg = G^T dU G.  16 multiplies per (n, k, tile) instead of 36 (9 taps x 4 pixels).
//   A workgroup owns BN output channels x BK input channels and a range of 16-tile chunks (split over the
//   tiles: partial slabs + the deterministic wgrad_reduce of conv.hip, same slab layout [tap][cout][cin]).
//   Per chunk the dY rows and the haloed X rows are staged global -> registers -> LDS; each wave (32 n x 16 k) then
//   runs 4 reduction steps of 4 tiles: lane (li, lq) transforms ITS operands in registers -- A dY A^T for rows
//   n = li, 16 + li and B^T d B for column k = li, both at tile 4s + lq (12 + 12 + 32 adds; the minus signs of
//   A dY A^T are applied once, in the epilogue) -- and issues 32 v_mfma_f32_16x16x4_f32.  No operand is read from
//   LDS inside the multiply loop.  The epilogue folds G^T dU G in registers (the 16 xi of an (n, k) pair sit in one
//   lane) and writes 9 taps.
// ------------------------------------------------------------------------------------------
template <int S> struct WgGeo {
  using G = WGeo<S, 16>;
  static constexpr int W = S, HW = S * S, Wp = G::Wp, PXR = G::PXR, TI = G::TI, RTX = PXR + 2, IMG = G::IMG, RS = G::RS;
  static constexpr int RSX = RS + ((4 - RS % 8) + 8) % 8;          // = 4 * odd: 16 channel rows land on 16 distinct bank quads
  static constexpr int DS = 68;                                     // dY: 64 floats per channel (16 tiles x 2 x 2), stride 4 * 17
  static constexpr int poff(int t) { return (t / (G::TROWS * G::TPR)) * IMG + 2 * ((t / G::TPR) % G::TROWS) * Wp + 2 * (t % G::TPR); }
  static constexpr int doff(int t) { return (t / (G::TROWS * G::TPR)) * PXR * W + 2 * ((t / G::TPR) % G::TROWS) * W + 2 * (t % G::TPR); }
};

template <int S, int BN, int BK>
__global__ __launch_bounds__((BN / 32) * (BK / 16) * 64, (BN / 32) * (BK / 16) == 8 ? 1 : 2)
void wgrad_wino(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                int B, int Cin, int Cout, int chunks_per_split, int nchunks) {
  using Q = WgGeo<S>;
  using G = WGeo<S, 16>;
  constexpr int W = S, H = S, HW = S * S, Wp = Q::Wp, PXR = Q::PXR, TI = Q::TI, RTX = Q::RTX, IMG = Q::IMG, RSX = Q::RSX, DS = Q::DS;
  constexpr int NWN = BN / 32, NWK = BK / 16, NTH = NWN * NWK * 64, W4 = W / 4;
  constexpr int XUNITS = BK * TI * RTX * W4, NX = (XUNITS + NTH - 1) / NTH;        // float4 pieces per chunk / per thread
  constexpr int DUNITS = BN * TI * PXR * W4, ND = (DUNITS + NTH - 1) / NTH;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                                  // [BK][RSX]  haloed input rows (halo columns stay zero)
  float* Gs = smem + BK * RSX;                       // [BN][DS]   dY rows
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n0 = blockIdx.x * BN, k0 = blockIdx.y * BK, split = blockIdx.z;
  const int wn = wv % NWN, wk = wv / NWN;
  const int li = lane & 15, lq = lane >> 4;

  for (int i = tid; i < BK * RSX; i += NTH) Xs[i] = 0.f;

  // chunk -> (first image, first pixel row): running state of the next chunk to fetch
  const int cbeg = split * chunks_per_split, cend = min(nchunks, cbeg + chunks_per_split);
  int f_b = (TI == 1) ? cbeg / G::GPI : cbeg * TI;
  int f_g = (TI == 1) ? cbeg % G::GPI : 0;

  f32x4 xreg[NX], greg[ND];
  unsigned xok, gok;
  auto fetch = [&]() {
    const int row0 = f_g * PXR;
    xok = 0; gok = 0;
#pragma unroll
    for (int e = 0; e < NX; ++e) {
      const int u = tid + NTH * e;
      const int x4 = u % W4, ch = (u / W4) % BK, rl = u / (W4 * BK);
      const int ti = rl / RTX, rr = rl % RTX;
      const int yy = row0 - 1 + rr, b = f_b + ti;
      const bool ok = (XUNITS % NTH == 0 || u < XUNITS) && yy >= 0 && yy < H && b < B;
      const unsigned src = ok ? (unsigned)((b * Cin + k0 + ch) * HW + yy * W + 4 * x4) : 0u;
      xreg[e] = *reinterpret_cast<const f32x4*>(x + src);
      xok |= (ok ? 1u : 0u) << e;
    }
#pragma unroll
    for (int e = 0; e < ND; ++e) {
      const int u = tid + NTH * e;
      const int x4 = u % W4, n = (u / W4) % BN, rl = u / (W4 * BN);
      const int ti = rl / PXR, rr = rl % PXR;
      const int b = f_b + ti;
      const bool ok = (DUNITS % NTH == 0 || u < DUNITS) && b < B;
      const unsigned src = ok ? (unsigned)((b * Cout + n0 + n) * HW + (row0 + rr) * W + 4 * x4) : 0u;
      greg[e] = *reinterpret_cast<const f32x4*>(dy + src);
      gok |= (ok ? 1u : 0u) << e;
    }
    if (TI == 1) { if (++f_g == G::GPI) { f_g = 0; ++f_b; } } else { f_b += TI; }
  };
  auto commit = [&]() {
#pragma unroll
    for (int e = 0; e < NX; ++e) {
      const int u = tid + NTH * e;
      if (XUNITS % NTH == 0 || u < XUNITS) {
        const int x4 = u % W4, ch = (u / W4) % BK, rl = u / (W4 * BK);
        float* d = Xs + ch * RSX + (rl / RTX) * IMG + (rl % RTX) * Wp + 1 + 4 * x4;
        const bool ok = (xok >> e) & 1u;
        d[0] = ok ? xreg[e][0] : 0.f; d[1] = ok ? xreg[e][1] : 0.f; d[2] = ok ? xreg[e][2] : 0.f; d[3] = ok ? xreg[e][3] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < ND; ++e) {
      const int u = tid + NTH * e;
      if (DUNITS % NTH == 0 || u < DUNITS) {
        const int x4 = u % W4, n = (u / W4) % BN, rl = u / (W4 * BN);
        const bool ok = (gok >> e) & 1u;
        f32x4 v = greg[e];
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(Gs + n * DS + (rl / PXR) * PXR * W + (rl % PXR) * W + 4 * x4) = v;
      }
    }
  };

  const float* xp = Xs + (wk * 16 + li) * RSX + Q::poff(0) + (lq == 0 ? Q::poff(0) : lq == 1 ? Q::poff(1) : lq == 2 ? Q::poff(2) : Q::poff(3));
  const float* gp = Gs + (wn * 32 + li) * DS + (lq == 0 ? Q::doff(0) : lq == 1 ? Q::doff(1) : lq == 2 ? Q::doff(2) : Q::doff(3));

  f32x4 acc[16][2];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[xi][h] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifndef AFD_WGW_ABL
#define AFD_WGW_ABL 0
#endif
  if (cbeg < cend) fetch();
  for (int c = cbeg; c < cend; ++c) {
    __syncthreads();                                  // the previous chunk's transforms are done with Xs / Gs (first pass: the zero fill)
    if (!(AFD_WGW_ABL & 8) || c == cbeg) commit();
    if (c + 1 < cend && (!(AFD_WGW_ABL & 4))) fetch();                        // in flight during the multiplies
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // B operand: V = B^T d B of (channel wk*16 + li, tile 4s + lq)
      float v[16];
      {
        const float* r = xp + Q::poff(4 * s);
        float d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const float2 lo = *reinterpret_cast<const float2*>(r + a * Wp);
          const float2 hi = *reinterpret_cast<const float2*>(r + a * Wp + 2);
          d[a][0] = lo.x; d[a][1] = lo.y; d[a][2] = hi.x; d[a][3] = hi.y;
        }
        float w[4][4];
        if (AFD_WGW_ABL & 1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = d[i >> 2][i & 3];
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          w[0][j] = d[0][j] - d[2][j];
          w[1][j] = d[1][j] + d[2][j];
          w[2][j] = d[2][j] - d[1][j];
          w[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[4 * i + 0] = w[i][0] - w[i][2];
          v[4 * i + 1] = w[i][1] + w[i][2];
          v[4 * i + 2] = w[i][2] - w[i][1];
          v[4 * i + 3] = w[i][1] - w[i][3];
        }
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // A operand: |A dY A^T| of (channel wn*32 + 16h + li, tile 4s + lq); entries 3, 7, 11, 12, 13, 14 carry a minus
        // sign that the epilogue applies
        const float* q = gp + h * 16 * DS + Q::doff(4 * s);
        const float2 r0 = *reinterpret_cast<const float2*>(q), r1 = *reinterpret_cast<const float2*>(q + W);
        const float a = r0.x, b = r0.y, cc = r1.x, d = r1.y;
        const float ac = a + cc, bd = b + d, amc = a - cc, bmd = b - d;
        float m[16];
        m[0] = a;    m[1] = a + b;     m[2] = a - b;     m[3] = b;
        m[4] = ac;   m[5] = ac + bd;   m[6] = ac - bd;   m[7] = bd;
        m[8] = amc;  m[9] = amc + bmd; m[10] = amc - bmd; m[11] = bmd;
        m[12] = cc;  m[13] = cc + d;   m[14] = cc - d;   m[15] = d;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
          if (AFD_WGW_ABL & 2) { asm volatile("" :: "v"(m[xi]), "v"(v[xi])); }
          else acc[xi][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[xi], v[xi], acc[xi][h], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: dg = G^T dU G with G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]; slab layout [tap][cout][cin]
  float* out = part + (long)split * Cout * Cin * 9;
  const long plane = (long)Cout * Cin;
  const int kk = k0 + wk * 16 + li;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned o = (unsigned)((n0 + wn * 32 + 16 * h + 4 * lq + r) * Cin + kk);     // lane offset inside a tap plane
      float t[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sg3 = (j == 3) ? 1.f : -1.f;               // row 3 of |A dY A^T|: -, -, -, +
        const float u0 = (j == 3) ? -acc[j][h][r] : acc[j][h][r];
        const float u1 = (j == 3) ? -acc[4 + j][h][r] : acc[4 + j][h][r];
        const float u2 = (j == 3) ? -acc[8 + j][h][r] : acc[8 + j][h][r];
        const float u3 = sg3 * acc[12 + j][h][r];
        const float hs = 0.5f * (u1 + u2), hd = 0.5f * (u1 - u2);
        t[0][j] = u0 + hs; t[1][j] = hd; t[2][j] = hs + u3;
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float hs = 0.5f * (t[a][1] + t[a][2]), hd = 0.5f * (t[a][1] - t[a][2]);
        (out + (3 * a + 0) * plane)[o] = t[a][0] + hs;
        (out + (3 * a + 1) * plane)[o] = hd;
        (out + (3 * a + 2) * plane)[o] = hs + t[a][3];
      }
    }
}

static int g_wgwino_mode = 0;      // 0 = by rule, 1 = off, 2 = whenever covered
void wgrad_wino_set_mode(int m) { g_wgwino_mode = m; }

// Winograd wgrad plan: returns the number of slabs (0 = not covered / not chosen); fills the block shape and split
int wgrad_wino_plan(int B, int Cin, int Cout, int H, int W, int* bn, int* bk, int* cps, int* nchunks) {
  if (g_wgwino_mode == 1) return 0;
  if (H != W || (W != 32 && W != 16 && W != 8 && W != 4)) return 0;
  if (Cin % 32 || Cout % 32) return 0;
  if ((long)B * H * W * (Cin > Cout ? Cin : Cout) >= (1L << 31)) return 0;
  const int tpi = (W / 2) * (W / 2);
  const int nch = tpi >= 16 ? B * (tpi / 16) : (B + 16 / tpi - 1) / (16 / tpi);
  const int BN = Cout % 64 == 0 ? 64 : 32, BK = Cin % 64 == 0 ? 64 : 32;
  const int waves = (BN / 32) * (BK / 16);
  const long blocks = (long)(Cout / BN) * (Cin / BK);
  static const long target = [] { const char* e = getenv("AFD_WGW_TARGET"); return e ? atol(e) : 128L; }();   // tuning hook; 128 measured best (see below)
  long s = (target * (waves == 8 ? 1 : (waves == 4 ? 2 : 4))) / blocks;
  if (s < 1) s = 1;
  if (s > nch) s = nch;
  const int c = (int)((nch + s - 1) / s);
  *bn = BN; *bk = BK; *cps = c; *nchunks = nch;
  return (nch + c - 1) / c;
}

template <int S, int BN, int BK>
static void wgrad_wino_launch_t(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int cps, int nch, int splits,
                                hipStream_t s) {
  using Q = WgGeo<S>;
  const size_t lds = sizeof(float) * ((size_t)BK * Q::RSX + (size_t)BN * Q::DS);
  (void)lds_opt_in(&wgrad_wino<S, BN, BK>, lds);                    // (> 64 KB of dynamic LDS: once per device and kernel)
  hipLaunchKernelGGL((wgrad_wino<S, BN, BK>), dim3(Cout / BN, Cin / BK, splits), dim3((BN / 32) * (BK / 16) * 64), lds, s,
                     x, dy, part, B, Cin, Cout, cps, nch);
}

// writes `slabs` partial [9][Cout][Cin] slabs into part; the caller reduces them (wgrad_reduce)
int wgrad_wino(const float* x, const float* dy, float* part, int B, int Cin, int Cout, int H, int W, hipStream_t s) {
  int bn, bk, cps, nch;
  const int splits = wgrad_wino_plan(B, Cin, Cout, H, W, &bn, &bk, &cps, &nch);
  if (!splits) return 0;
#define AFD_WGW(S_)                                                                                              \
  if (bn == 64 && bk == 64) wgrad_wino_launch_t<S_, 64, 64>(x, dy, part, B, Cin, Cout, cps, nch, splits, s);     \
  else if (bn == 32 && bk == 64) wgrad_wino_launch_t<S_, 32, 64>(x, dy, part, B, Cin, Cout, cps, nch, splits, s); \
  else if (bn == 64) wgrad_wino_launch_t<S_, 64, 32>(x, dy, part, B, Cin, Cout, cps, nch, splits, s);            \
  else wgrad_wino_launch_t<S_, 32, 32>(x, dy, part, B, Cin, Cout, cps, nch, splits, s)
  if (W == 32) { AFD_WGW(32); } else if (W == 16) { AFD_WGW(16); } else if (W == 8) { AFD_WGW(8); } else { AFD_WGW(4); }
#undef AFD_WGW
  return splits;
}

// both (or either) transformed-weight images of a layer; kinds bit 0 / 1: the forward / dgrad image is the direct form's
// (f16x2; with bit 2: bf16x3)
void wino_weights_launch(const float* w, float* Uf, float* Ud, int Cin, int Cout, int kinds, hipStream_t s) {
  float* wf = (kinds & 1) ? nullptr : Uf; float* wd = (kinds & 2) ? nullptr : Ud;
  if (wf || wd) hipLaunchKernelGGL(wino_weights, dim3((unsigned)((Cin * Cout / 64 + 3) / 4)), dim3(256), 0, s, w, wf, wd, Cin, Cout);
  void* bf = (kinds & 1) ? Uf : nullptr; void* bd = (kinds & 2) ? Ud : nullptr;
  if (bf || bd) {
    if (kinds & 4) bf3_weights_launch(w, bf, bd, Cin, Cout, s);
    else h2_weights_launch(w, bf, bd, Cin, Cout, s);
  }
}

void wino_weights_batched_launch(const afd_wino_desc* descs, const int* wg_desc, int n_wg, hipStream_t s) {
  hipLaunchKernelGGL(h2_wscale_batched, dim3((unsigned)n_wg), dim3(256), 0, s, descs, wg_desc);
  hipLaunchKernelGGL(wino_weights_batched, dim3((unsigned)n_wg), dim3(256), 0, s, descs, wg_desc);
}

}  // namespace afd

// fold.hip -- the deterministic tail of every parameter gradient, batched.
//
// Every parameter-gradient producer of the step (3x3 / 1x1 weight-gradient slabs [split][tap][cout][cin], bias slabs,
// GroupNorm / LayerNorm (B, 2, C) plane partials) ends in the same operation: out[map(j)] (+)= sum over the partial slabs of
// element j, in a fixed order.  Round 2 launched one small kernel per producer for it (57 wgrad_reduce + 26 colsum2 launches
// per step, 3-15 us each, latency not bandwidth); here ONE launch folds up to kFoldMax of them: a fold is a 56-byte
// descriptor, the descriptors travel as kernel arguments (no device table, no H2D copy, legal under stream capture), a
// workgroup finds its descriptor by its block index.  No atomics; the summation order of an element depends on its
// descriptor alone, so results are bit-identical however the folds are batched.
//
// Element j of a slab lands at dst[(j % inner) * rstride + j / inner]:
//   identity            inner = n,            rstride = 1
//   [tap][plane] slabs  inner = plane = n/T,  rstride = T   (the weight gradient is OIHW: (cout*Cin+cin)*T + tap)
#include "common.h"

namespace afd {

constexpr int kFoldMax = 56;                    // 56 x 56 B + 57 x 4 B + 4 B = 3368 B of kernel arguments (limit 4096)

struct FoldArgs {
  afd_fold_desc d[kFoldMax];
  int first[kFoldMax + 1];                      // first workgroup of descriptor i; first[n] = grid size
  int n;
};

__device__ __forceinline__ bool fold_vec_ok(const afd_fold_desc& d) {
  return d.n % 128 == 0 && d.inner % 4 == 0 && d.stride % 4 == 0 && (reinterpret_cast<uintptr_t>(d.part) & 15) == 0;
}
static inline bool fold_vec_ok_host(const afd_fold_desc& d) {
  return d.n % 128 == 0 && d.inner % 4 == 0 && d.stride % 4 == 0 && (reinterpret_cast<uintptr_t>(d.part) & 15) == 0;
}

// 256 threads = 32 lanes x 8 slab groups.  Vector form: a workgroup owns 128 consecutive elements, every half-wave
// streams a 512-byte run of one slab (16-byte loads); scalar form: 32 consecutive elements.  Group g takes slabs
// g, g+8, ... into four interleaved running sums (4 loads in flight), the 8 group sums meet in LDS in group order.
__global__ __launch_bounds__(256) void fold_batched_k(const FoldArgs a) {
  __shared__ float4 red4[8][32];
  int i = 0;
  for (int k = 1; k < a.n; ++k) if ((int)blockIdx.x >= a.first[k]) i = k;           // uniform: scalar loop over kernel arguments
  const afd_fold_desc& d = a.d[i];
  const int blk = blockIdx.x - a.first[i];
  const int l = threadIdx.x & 31, g = threadIdx.x >> 5;
  const float* __restrict__ part = d.part;
  float* __restrict__ dst = d.dst;
  const long n = d.n, stride = d.stride, inner = d.inner, rstride = d.rstride;
  const int splits = d.splits, acc = d.accumulate;
  if (fold_vec_ok(d)) {
    const long j = ((long)blk << 7) + 4 * l;
    const float4* __restrict__ src = reinterpret_cast<const float4*>(part + j);
    const long s4 = stride >> 2;
    float4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int k = g;
    for (; k + 24 < splits; k += 32) {
      const float4 x0 = src[(long)k * s4], x1 = src[(long)(k + 8) * s4], x2 = src[(long)(k + 16) * s4], x3 = src[(long)(k + 24) * s4];
      s0.x += x0.x; s0.y += x0.y; s0.z += x0.z; s0.w += x0.w;
      s1.x += x1.x; s1.y += x1.y; s1.z += x1.z; s1.w += x1.w;
      s2.x += x2.x; s2.y += x2.y; s2.z += x2.z; s2.w += x2.w;
      s3.x += x3.x; s3.y += x3.y; s3.z += x3.z; s3.w += x3.w;
    }
    for (; k < splits; k += 8) {
      const float4 x0 = src[(long)k * s4];
      s0.x += x0.x; s0.y += x0.y; s0.z += x0.z; s0.w += x0.w;
    }
    red4[g][l] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                             (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (threadIdx.x < 128) {                                  // one element per thread: (float4 slot e >> 2, component e & 3)
      const int e = threadIdx.x;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += reinterpret_cast<const float*>(&red4[q][e >> 2])[e & 3];
      const long je = ((long)blk << 7) + e;
      float* o = dst + (je % inner) * rstride + je / inner;
      *o = acc ? *o + t : t;
    }
  } else {
    float (*red)[33] = reinterpret_cast<float (*)[33]>(&red4[0][0]);       // 8 x 33 floats inside the 4-KB buffer
    const long j = (long)blk * 32 + l;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < n) {
      int k = g;
      for (; k + 24 < splits; k += 32) {
        s0 += part[(long)k * stride + j]; s1 += part[(long)(k + 8) * stride + j];
        s2 += part[(long)(k + 16) * stride + j]; s3 += part[(long)(k + 24) * stride + j];
      }
      for (; k < splits; k += 8) s0 += part[(long)k * stride + j];
    }
    red[g][l] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && j < n) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += red[q][l];
      float* o = dst + (j % inner) * rstride + j / inner;
      *o = acc ? *o + t : t;
    }
  }
}

// host side, shared with conv.hip (the non-deferred weight-gradient entry point folds through the same kernel, so a
// deferred and an immediate fold of the same slabs give the same bits)
int fold_launch(const afd_fold_desc* descs, int n, hipStream_t s) {
  for (int base = 0; base < n; base += kFoldMax) {
    FoldArgs a;
    a.n = n - base < kFoldMax ? n - base : kFoldMax;
    long wg = 0;
    for (int i = 0; i < a.n; ++i) {
      const afd_fold_desc& d = descs[base + i];
      if (!(d.part && d.dst && d.n > 0 && d.splits > 0 && d.stride >= d.n && d.inner > 0 && d.n % d.inner == 0 && d.rstride > 0))
        return set_error(AFD_EINVAL, "afd_fold_batched: bad descriptor %d (n %ld, splits %d, stride %ld, inner %ld, rstride %ld)", base + i,
                         d.n, d.splits, d.stride, d.inner, d.rstride);
      a.d[i] = d;
      a.first[i] = (int)wg;
      wg += fold_vec_ok_host(d) ? d.n / 128 : (d.n + 31) / 32;
      if (wg > (1L << 30)) return set_error(AFD_EINVAL, "afd_fold_batched: grid too large");
    }
    for (int i = a.n; i <= kFoldMax; ++i) a.first[i] = (int)wg;
    hipLaunchKernelGGL(fold_batched_k, dim3((unsigned)wg), dim3(256), 0, s, a);
  }
  return AFD_OK;
}

}  // namespace afd
using namespace afd;

extern "C" {

int afd_fold_batched(const afd_fold_desc* descs, int n, afd_stream_t st) {
  AFD_REQUIRE(descs && n > 0, "afd_fold_batched: bad argument");
  const int rc = fold_launch(descs, n, as_stream(st));
  return rc != AFD_OK ? rc : check_launch("afd_fold_batched");
}

}  // extern "C"

// Ablation harness for conv_bf3 (csrc/bf3.hip): -DBF3_ABL=<bits>, times the forward of a few Config-D layers at B = 256.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DBF3_ABL=0 tools/micro/bf3_abl.hip \
//         aliasfree-diffusion-models-pytorch_amd/csrc/host.cpp -o tools/micro/bin/bf3_abl_0
#include "../../aliasfree-diffusion-models-pytorch_amd/csrc/bf3.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main() {
  const int B = 256;
  const int shapes[][3] = {{128, 128, 16}, {256, 256, 8}, {64, 64, 32}, {64, 64, 16}, {256, 128, 8}, {128, 256, 8}, {128, 128, 8},
                           {128, 64, 16}, {64, 128, 16}, {64, 32, 32}, {32, 64, 32}, {32, 32, 32}};
  const int nblk = getenv("BF3_NBLK") ? atoi(getenv("BF3_NBLK")) : 0;       // 0: the library's rule
  afd::bf3_set_nblk(nblk);
  for (auto& sh : shapes) {
    const int K = sh[0], N = sh[1], S = sh[2];
    const size_t nx = (size_t)B * K * S * S, ny = (size_t)B * N * S * S;
    float *x, *y, *w; void* wp;
    hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&w, (size_t)K * N * 9 * 4); hipMalloc(&wp, (size_t)64 * K * N);
    std::vector<float> h(nx);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), (size_t)K * N * 9 * 4, hipMemcpyHostToDevice);
    afd::bf3_weights_launch(w, wp, nullptr, K, N, nullptr);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) afd::bf3_conv(x, wp, nullptr, nullptr, y, B, K, N, S, 0, nullptr);
    hipEventRecord(a, nullptr);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) afd::bf3_conv(x, wp, nullptr, nullptr, y, B, K, N, S, 0, nullptr);
    hipEventRecord(b, nullptr); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double fl = 2.0 * B * S * S * (double)K * N * 9;
    printf("abl %2d nblk %d  %3d->%3d @%2dx%-2d  %7.1f us  %6.1f TF algorithmic\n", BF3_ABL, nblk, K, N, S, S, ms / reps * 1e3, fl / (ms / reps * 1e-3) / 1e12);
    hipFree(x); hipFree(y); hipFree(w); hipFree(wp);
  }
  return 0;
}

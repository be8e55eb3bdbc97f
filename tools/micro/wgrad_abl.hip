// Ablation harness for wgrad_wino (csrc/wino.hip): builds the kernel file itself with -DAFD_WGW_ABL=<bits> and times
// one layer.  bits: 1 = no operand transforms (raw patch values), 2 = no MFMAs, 4 = no global fetch after the first chunk,
// 8 = no LDS commit after the first chunk, 16 = no epilogue stores.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DAFD_WGW_ABL=0 tools/micro/wgrad_abl.hip \
//         aliasfree-diffusion-models-pytorch_amd/csrc/host.cpp -o tools/micro/bin/wgrad_abl_0
#include "../../aliasfree-diffusion-models-pytorch_amd/csrc/wino.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int B = 256;
  const int shapes[][3] = {{64, 64, 32}, {128, 128, 16}, {256, 256, 8}, {256, 256, 4}, {64, 32, 32}, {128, 128, 4}};
  for (auto& sh : shapes) {
    const int Cin = sh[0], Cout = sh[1], S = sh[2];
    const size_t nx = (size_t)B * Cin * S * S, ny = (size_t)B * Cout * S * S;
    float *x, *dy, *part;
    hipMalloc(&x, nx * 4); hipMalloc(&dy, ny * 4);
    std::vector<float> h(nx > ny ? nx : ny);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, h.data(), ny * 4, hipMemcpyHostToDevice);
    int bn, bk, cps, nch;
    const int splits = afd::wgrad_wino_plan(B, Cin, Cout, S, S, &bn, &bk, &cps, &nch);
    hipMalloc(&part, (size_t)splits * 9 * Cin * Cout * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) afd::wgrad_wino(x, dy, part, B, Cin, Cout, S, S, nullptr);
    hipEventRecord(a, nullptr);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) afd::wgrad_wino(x, dy, part, B, Cin, Cout, S, S, nullptr);
    hipEventRecord(b, nullptr); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double fl = 2.0 * B * S * S * (double)Cin * Cout * 9;
    printf("abl %2d  %3d->%3d @%2dx%-2d  splits %3d (chunks/split %d)  %7.1f us  %6.1f TF algorithmic\n", AFD_WGW_ABL, Cin, Cout, S, S, splits, cps,
           ms / reps * 1e3, fl / (ms / reps * 1e-3) / 1e12);
    hipFree(x); hipFree(dy); hipFree(part);
  }
  return 0;
}

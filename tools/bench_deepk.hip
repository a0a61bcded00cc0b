// Diagnostic build of the weights-stationary deep conv kernel (csrc/conv_deepk.hip) with in-kernel phase stamps:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDEVIS_PHASE_TIMING -Idevis_amd/csrc -Iinclude tools/bench_deepk.hip -o tools/bench_deepk
// Prints, per layer shape, the kernel time (HIP events, 20 launches) and the median over blocks of the shader-clock
// cycles each phase took: x loads + weights issue + staging | barrier | MFMA loop | partial-sum exchange | epilogue.
// Never part of the library.
#include "../devis_amd/csrc/conv_deepk.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
namespace devis {
thread_local int g_last_hip_error = 0;
}
using namespace devis;

template <int CIN, int TD, int TH>
static void run(int N, int S) {
  const size_t vox = (size_t)N * S * S * S;
  std::vector<unsigned short> hx(vox * CIN), hw((size_t)CIN * 27 * CIN);
  unsigned r = 12345u;
  auto rnd = [&]() { r = r * 1664525u + 1013904223u; return (unsigned short)(0x3c00u + ((r >> 9) & 0x3ff) - ((r >> 20) & 1) * 0x8000u * 0); };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = (unsigned short)(rnd() - 0x0400u);
  bf16 *x, *w, *y; long long* dbg;
  hipMalloc(&x, hx.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&y, hx.size() * 2);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  DeepKArgs a{};
  a.x = x; a.w = w; a.bias = nullptr; a.sc = a.sh = nullptr; a.res = nullptr; a.y = y;
  a.N = N; a.D = a.H = a.W = S; a.Cout = CIN;
  a.ntd = (S + TD - 1) / TD; a.nth = (S + TH - 1) / TH; a.ntw = (S + 15) / 16;
  a.st = StatsArgs{};
  const int tiles = N * a.ntd * a.nth * a.ntw, gy = CIN / (32 * DeepKCfg<CIN, TD, TH>::MG);
  hipMalloc(&dbg, (size_t)tiles * gy * 8 * sizeof(long long));
  a.dbg = dbg;
  auto k = k_conv3_deepk<CIN, TD, TH, false, 0>;
  constexpr int lds = DeepKCfg<CIN, TD, TH>::LDS;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles, gy), dim3(256), lds, 0, a);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(tiles, gy), dim3(256), lds, 0, a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h((size_t)tiles * gy * 8);
  hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
  const int nb = tiles * gy;
  printf("%3d ch @ %d^3 x %d: %d blocks, %.1f us per launch (events, back to back)\n", CIN, S, N, nb, ms / 20 * 1e3);
  const char* names[5] = {"loads+stage", "barrier", "mfma loop", "exchange", "epilogue"};
  for (int p = 0; p < 5; ++p) {
    std::vector<long long> d(nb);
    for (int b = 0; b < nb; ++b) d[b] = h[(size_t)b * 8 + p + 1] - h[(size_t)b * 8 + p];
    std::sort(d.begin(), d.end());
    printf("   %-12s median %7lld  min %7lld  max %7lld shader cycles\n", names[p], d[nb / 2], d[0], d[nb - 1]);
  }
  {
    std::vector<long long> d(nb);
    for (int b = 0; b < nb; ++b) d[b] = h[(size_t)b * 8 + 6] - h[(size_t)b * 8 + 0];
    std::sort(d.begin(), d.end());
    printf("   (of loads+stage: issuing the tile loads and the first weight loads took median %lld cycles)\n", d[nb / 2]);
  }
  long long rmin = h[7], rmax = h[7];
  for (int b = 0; b < nb; ++b) { rmin = std::min(rmin, h[(size_t)b * 8 + 7]); rmax = std::max(rmax, h[(size_t)b * 8 + 7]); }
  printf("   block start spread %.2f us (100 MHz realtime)\n", (rmax - rmin) / 100.0);
  hipFree(x); hipFree(w); hipFree(y); hipFree(dbg);
}

int main() {
  run<64, 4, 4>(2, 32);
  run<128, 2, 4>(2, 16);
  return 0;
}

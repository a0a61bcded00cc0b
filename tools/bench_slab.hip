// Phase timing of k_wgrad3_slab_march on one block (developer tool, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDEVIS_PHASE_TIMING -Idevis_amd/csrc -Iinclude tools/bench_slab.hip -o tools/bench_slab
#include "../devis_amd/csrc/conv_wgrad_march.hip"
#include <stdio.h>
#include <vector>
namespace devis {
thread_local int g_last_hip_error = 0;
int launch_wgrad_reduce(const float*, const float*, int, int, int, int, float*, float*, hipStream_t) { return 0; }
int launch_wgrad_reduce_frag(const float*, const float*, int, int, int, float*, float*, hipStream_t) { return 0; }
}
static void run(int N, int S, int C) {
  const size_t elems = (size_t)N * S * S * S * C;
  void *x, *dy; float *sc, *sh, *dw, *db; void* ws;
  hipMalloc(&x, elems * 2); hipMalloc(&dy, elems * 2);
  std::vector<unsigned short> h16(elems);
  unsigned r = 7u;
  for (auto& v : h16) { r = r * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((r >> 9) & 0x3ff) + ((r >> 3) & 0x8000u)); }
  hipMemcpy(x, h16.data(), elems * 2, hipMemcpyHostToDevice); hipMemcpy(dy, h16.data(), elems * 2, hipMemcpyHostToDevice);
  hipMalloc(&sc, N * C * 4); hipMalloc(&sh, N * C * 4);
  std::vector<float> one(N * C, 1.f); hipMemcpy(sc, one.data(), N * C * 4, hipMemcpyHostToDevice); hipMemset(sh, 0, N * C * 4);
  hipMalloc(&dw, 27 * C * C * 4); hipMalloc(&db, 4096);
  devis_conv_desc d = {N, S, S, S, S, S, S, C, 0, C, 3, 3, 3, 1, 1, 1, 1, 1, 1, DEVIS_BF16};
  size_t need = 0;
  devis::wgrad3_slab_march(&d, x, sc, sh, dy, dw, db, nullptr, 0, 0, true, &need);
  hipMalloc(&ws, need + 4096);
  for (int rep = 0; rep < 3; ++rep) devis::wgrad3_slab_march(&d, x, sc, sh, dy, dw, db, ws, need + 4096, 0, false, &need);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int rep = 0; rep < 20; ++rep) devis::wgrad3_slab_march(&d, x, sc, sh, dy, dw, db, ws, need + 4096, 0, false, &need);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // the kernel wrote its phase counters over the start of the bias partials
  const size_t gx = need / ((size_t)27 * C * C * 4 + C * 4);
  std::vector<long long> h(64);
  hipMemcpy(h.data(), (char*)ws + need, 64 * 8, hipMemcpyDeviceToHost);
  printf("%3d ch @ %d^3 x %d: %.1f us per launch; partials %zu x %.0f KB = %.1f MB; cycles of block 0 (issue loads | fragments+MFMA | affine+LDS store | barrier):\n",
         C, S, N, ms / 20 * 1e3, gx, 27.0 * C * C * 4 / 1024, gx * 27.0 * C * C * 4 / 1e6);
  for (int w = 0; w < 8; ++w) printf("  wave %d (%s): %lld %lld %lld %lld | fill %lld, partial store %lld, bias %lld\n", w, w < 4 ? "mfma" : "load", h[w * 8], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 5], h[w * 8 + 6]);
  hipFree(x); hipFree(dy); hipFree(sc); hipFree(sh); hipFree(dw); hipFree(db); hipFree(ws);
}
int main() {
  run(2, 64, 32);
  run(2, 32, 64);
  run(2, 16, 128);
  return 0;
}

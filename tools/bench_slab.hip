// Phase timing of k_wgrad3_slab_march on one block (developer tool, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDEVIS_PHASE_TIMING tools/bench_slab.hip -o tools/bench_slab
#include "../devis_amd/csrc/conv_wgrad_march.hip"
#include <stdio.h>
#include <vector>
namespace devis {
thread_local int g_last_hip_error = 0;
int launch_wgrad_reduce(const float*, const float*, int, int, int, int, float*, float*, hipStream_t) { return 0; }
}
int main() {
  const int N = 2, S = 64, C = 32;
  const size_t elems = (size_t)N * S * S * S * C;
  void *x, *dy; float *sc, *sh, *dw, *db; void* ws;
  hipMalloc(&x, elems * 2); hipMalloc(&dy, elems * 2);
  hipMemset(x, 0x3c, elems * 2); hipMemset(dy, 0x3c, elems * 2);
  hipMalloc(&sc, N * C * 4); hipMalloc(&sh, N * C * 4); hipMemset(sc, 0, N * C * 4); hipMemset(sh, 0, N * C * 4);
  hipMalloc(&dw, 27 * C * C * 4); hipMalloc(&db, 4096);
  devis_conv_desc d = {N, S, S, S, S, S, S, C, 0, C, 3, 3, 3, 1, 1, 1, 1, 1, 1, DEVIS_BF16};
  size_t need = 0;
  devis::wgrad3_slab_march(&d, x, sc, sh, dy, dw, db, nullptr, 0, 0, true, &need);
  hipMalloc(&ws, need + 4096);
  for (int rep = 0; rep < 3; ++rep) devis::wgrad3_slab_march(&d, x, sc, sh, dy, dw, db, ws, need + 4096, 0, false, &need);
  hipDeviceSynchronize();
  // the kernel wrote its phase counters over the start of the bias partials
  const size_t gx = need / (27 * C * C * 4 + C * 4);
  std::vector<long long> h(32);
  hipMemcpy(h.data(), (char*)ws + need, 32 * 8, hipMemcpyDeviceToHost);
  printf("blocks %zu; cycles per walk of block 0 (issue loads | fragments+MFMA | affine+LDS store | barrier):\n", gx);
  for (int w = 0; w < 4; ++w) printf("  wave %d: %lld %lld %lld %lld | fill %lld, partial store %lld, bias %lld\n", w, h[w * 8], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 5], h[w * 8 + 6]);
  return 0;
}

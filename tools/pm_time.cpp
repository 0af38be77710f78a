// Phase timing of token_projmlp (debug aid, not part of the product): HAT geometry, 65 536 tokens x 180, hidden 360.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DTM_TIMING -Iimage-super-resolution-2_amd/csrc tools/pm_time.cpp -o tools/_dbg/pm_time
#include <stdarg.h>
#include <vector>
#include "../image-super-resolution-2_amd/csrc/token_mlp.hip"

static char g_err[512];
extern "C" const char* ff_last_error(void) { return g_err; }
void ff_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }

int main(int argc, char** argv) {
  const int nterms = argc > 1 ? atoi(argv[1]) : 1, with_c2 = argc > 2 ? atoi(argv[2]) : 1;
  const long long M = 65536; const int C = 180, HT = 12;
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  std::vector<float> h(M * C); for (auto& v : h) v = 2.f * rnd();
  auto bf = [&](size_t n, float sc) { std::vector<unsigned short> w(n); for (auto& v : w) { float f = sc * rnd(); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); } return w; };
  auto wp = bf((size_t)6 * 2 * 6144, 0.15f), wm = bf((size_t)HT * 4 * 6144, 0.15f);
  float *att, *x, *c2, *out, *vec; void *dwp, *dwm; unsigned long long* dbg;
  hipMalloc(&att, M * C * 4); hipMalloc(&x, M * C * 4); hipMalloc(&c2, M * C * 4); hipMalloc(&out, M * C * 4); hipMalloc(&vec, 4096 * 4);
  hipMalloc(&dwp, wp.size() * 2); hipMalloc(&dwm, wm.size() * 2);
  const size_t ndbg = 256 * 8 * 64; hipMalloc(&dbg, ndbg * 8);
  hipMemcpy(att, h.data(), M * C * 4, hipMemcpyHostToDevice); hipMemcpy(x, h.data(), M * C * 4, hipMemcpyHostToDevice);
  hipMemcpy(c2, h.data(), M * C * 4, hipMemcpyHostToDevice);
  std::vector<float> hv(4096, 0.01f); hipMemcpy(vec, hv.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dwp, wp.data(), wp.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dwm, wm.data(), wm.size() * 2, hipMemcpyHostToDevice);
  auto run = [&]() {
    return ff_token_projmlp(att, C, x, C, with_c2 ? c2 : nullptr, C, with_c2 ? vec : nullptr, out, C, M, C, HT, dwp, vec + 256, vec + 512, vec + 768, 1e-5f,
                            dwm, vec + 1024, vec + 2048, nterms, 0, nullptr);
  };
  g_tm_dbg = nullptr;
  for (int i = 0; i < 3; ++i) if (run()) { printf("error: %s\n", g_err); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int i = 0; i < 20; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("nterms %d c2 %d: %.1f us per launch (no stamps)\n", nterms, with_c2, ms * 1000 / 20);
  g_tm_dbg = dbg; hipMemset(dbg, 0, ndbg * 8);
  run(); hipDeviceSynchronize();
  std::vector<unsigned long long> t(ndbg);
  hipMemcpy(t.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull;
  for (size_t i = 0; i < ndbg; i += 64) if (t[i] && t[i] < t0) t0 = t[i];
  const char* names[64] = {};
  names[0] = "start"; names[1] = "att gathered"; names[2] = "proj done"; names[3] = "barrier"; names[4] = "residuals added"; names[5] = "LN done";
  names[6] = "barrier"; names[40] = "MLP done"; names[41] = "stores issued"; names[42] = "stores drained";
  static char nb[64][40];
  const char* ph[4] = {"gemm1 done", "barrier", "gelu+gemm2 done", "barrier"};
  for (int ht = 0; ht < 3; ++ht) for (int k = 0; k < 4; ++k) { snprintf(nb[8 + 4 * ht + k], 40, "ht%d %s", ht, ph[k]); names[8 + 4 * ht + k] = nb[8 + 4 * ht + k]; }
  double prev = 0;
  for (int i = 0; i < 64; ++i) {
    if (!names[i]) continue;
    double sum = 0, mn = 1e30, mx = 0; int n = 0;
    for (size_t w8 = 0; w8 < 256 * 8; ++w8) { const unsigned long long v = t[w8 * 64 + i]; if (!v) continue; const double u = (v - t0) / 100.0; sum += u; if (u < mn) mn = u; if (u > mx) mx = u; ++n; }
    if (!n) continue;
    printf("%-22s mean %7.2f us  (min %7.2f max %7.2f)  +%6.2f\n", names[i], sum / n, mn, mx, sum / n - prev);
    prev = sum / n;
  }
  return 0;
}

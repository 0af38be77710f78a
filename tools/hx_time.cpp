// Phase timing of the LDS-resident 3x3 convolution (debug aid, not part of the product): HAT's CAB geometry, 256 x 256 pixels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DHX_TIMING -Iimage-super-resolution-2_amd/csrc tools/hx_time.cpp -o tools/_dbg/hx_time
//   tools/_dbg/hx_time <Cin> <Cout> <bn> [nterms = 1] [pool = 1]
#include <stdarg.h>
#include <vector>
#include "../image-super-resolution-2_amd/csrc/conv3x3_halo.hip"

static char g_err[512];
extern "C" const char* ff_last_error(void) { return g_err; }
void ff_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }

int main(int argc, char** argv) {
  const int Cin = argc > 1 ? atoi(argv[1]) : 60, Cout = argc > 2 ? atoi(argv[2]) : 180, bn = argc > 3 ? atoi(argv[3]) : 192;
  const int nterms = argc > 4 ? atoi(argv[4]) : 1, pool = argc > 5 ? atoi(argv[5]) : 1;
  const int H = 256, W = 256;
  const long long M = (long long)H * W;
  unsigned s = 4242;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  std::vector<float> hx(M * Cin); for (auto& v : hx) v = rnd();
  const long long wb = ff_conv3x3_halo_weight_bytes(Cout, Cin, bn, nterms);
  std::vector<unsigned short> hw(wb / 2); for (auto& v : hw) { float f = 0.05f * rnd(); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  float *x, *out, *bias, *part; void* w; unsigned long long* dbg;
  hipMalloc(&x, M * Cin * 4); hipMalloc(&out, M * 192 * 4); hipMalloc(&bias, 256 * 4); hipMalloc(&w, wb);
  const long long prow = ff_conv3x3_halo_pool_rows(1, H, W, Cout, bn, nterms);
  hipMalloc(&part, prow * bn * 4);
  const size_t ndbg = 4096 * 8 * 8; hipMalloc(&dbg, ndbg * 8);
  hipMemcpy(x, hx.data(), M * Cin * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), wb, hipMemcpyHostToDevice); hipMemset(bias, 0, 1024);
  auto run = [&]() { return ff_conv3x3_halo(x, Cin, w, bn, bias, nullptr, nullptr, 0, out, Cout, 1, H, W, Cin, Cout, Cout <= 64 ? 1 : 0, 1.f, 0, (pool && Cout <= bn) ? part : nullptr, nterms, 0, nullptr); };
  g_hx_dbg = nullptr;
  for (int i = 0; i < 3; ++i) if (run()) { printf("error: %s\n", g_err); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int i = 0; i < 20; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("conv3x3 %d -> %d (bn %d, nterms %d, %lld workgroups): %.1f us per launch\n", Cin, Cout, bn, nterms, prow, ms * 1000 / 20);
  g_hx_dbg = dbg; hipMemset(dbg, 0, ndbg * 8);
  run(); hipDeviceSynchronize();
  std::vector<unsigned long long> t(ndbg);
  hipMemcpy(t.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull;
  for (size_t i = 0; i < ndbg; i += 8) if (t[i] && t[i] < t0) t0 = t[i];
  const char* names[5] = {"start", "input tile staged", "taps done", "stores issued", "stores drained"};
  double prev[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 5; ++i) {
    double sum = 0, mn = 1e30, mx = 0, dsum = 0; int n = 0;
    for (size_t w8 = 0; w8 < 4096 * 8; ++w8) { const unsigned long long v = t[w8 * 8 + i]; if (!v) continue; const double u = (v - t0) / 100.0; sum += u; if (u < mn) mn = u; if (u > mx) mx = u; if (i) dsum += (v - t[w8 * 8 + i - 1]) / 100.0; ++n; }
    if (!n) continue;
    printf("%-20s mean %7.2f us (min %7.2f max %7.2f)   phase mean %6.2f us   [%d waves]\n", names[i], sum / n, mn, mx, i ? dsum / n : 0.0, n);
  }
  return 0;
}

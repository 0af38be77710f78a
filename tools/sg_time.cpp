// Phase timing of the DAT SGFN tail (debug aid, not part of the product): 256 x 256 tokens, hidden 360, 180 outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DSG_TIMING -Iimage-super-resolution-2_amd/csrc tools/sg_time.cpp -o tools/_dbg/sg_time
#include <stdarg.h>
#include <vector>
#include "../image-super-resolution-2_amd/csrc/sgfn_tail.hip"

static char g_err[512];
extern "C" const char* ff_last_error(void) { return g_err; }
void ff_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }

int main() {
  const int H = 256, W = 256, c2 = 360, N = 180, HT = 12, ldh = 736;
  const long long M = (long long)H * W;
  unsigned s = 99;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  std::vector<float> hh(M * ldh); for (auto& v : hh) v = rnd();
  std::vector<float> hs(M * 2); for (size_t i = 0; i < hs.size(); i += 2) { hs[i] = 0.1f; hs[i + 1] = 1.5f; }
  std::vector<unsigned short> hw((size_t)HT * 192 * 32); for (auto& v : hw) { float f = 0.1f * rnd(); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  float *h, *st, *vec, *res, *out; void* w; unsigned long long* dbg;
  hipMalloc(&h, M * ldh * 4); hipMalloc(&st, M * 8); hipMalloc(&vec, 8192 * 4); hipMalloc(&res, M * 192 * 4); hipMalloc(&out, M * 192 * 4); hipMalloc(&w, hw.size() * 2);
  const size_t ndbg = 256 * 8 * 8; hipMalloc(&dbg, ndbg * 8);
  hipMemcpy(h, hh.data(), M * ldh * 4, hipMemcpyHostToDevice); hipMemcpy(st, hs.data(), M * 8, hipMemcpyHostToDevice);
  std::vector<float> hv(8192, 0.05f); hipMemcpy(vec, hv.data(), 8192 * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemset(res, 0, M * 192 * 4);
  auto run = [&]() { return ff_sgfn_tail(h, ldh, c2, st, vec, vec + 512, vec + 1024, vec + 5000, w, HT, vec + 6000, res, 192, out, 192, 1, H, W, N, nullptr); };
  g_sg_dbg = nullptr;
  for (int i = 0; i < 3; ++i) if (run()) { printf("error: %s\n", g_err); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int i = 0; i < 20; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("sgfn_tail: %.1f us per launch\n", ms * 1000 / 20);
  g_sg_dbg = dbg; hipMemset(dbg, 0, ndbg * 8);
  run(); hipDeviceSynchronize();
  std::vector<unsigned long long> t(ndbg);
  hipMemcpy(t.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull;
  for (size_t i = 0; i < ndbg; i += 8) if (t[i] && t[i] < t0) t0 = t[i];
  const char* names[6] = {"start", "tables + first weight tile", "chunk 0 staged", "12 chunks done", "stores issued", "stores drained"};
  for (int i = 0; i < 6; ++i) {
    double sum = 0, mn = 1e30, mx = 0, dsum = 0; int n = 0;
    for (size_t w8 = 0; w8 < 256 * 8; ++w8) { const unsigned long long v = t[w8 * 8 + i]; if (!v) continue; const double u = (v - t0) / 100.0; sum += u; if (u < mn) mn = u; if (u > mx) mx = u; if (i) dsum += (v - t[w8 * 8 + i - 1]) / 100.0; ++n; }
    if (n) printf("%-28s mean %7.2f us (min %7.2f max %7.2f)   phase mean %6.2f us\n", names[i], sum / n, mn, mx, i ? dsum / n : 0.0);
  }
  return 0;
}

// Phase timing of the persistent OCAB attention kernel (debug aid, not part of the product): 256 x 256 tokens, 6 heads.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DOC_TIMING -Iimage-super-resolution-2_amd/csrc tools/oc_time.cpp -o tools/_dbg/oc_time
#include <stdarg.h>
#include <vector>
#include "../image-super-resolution-2_amd/csrc/ocab_attn.hip"
static char g_err[512];
extern "C" const char* ff_last_error(void) { return g_err; }
void ff_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
int main() {
  const int H = 256, W = 256, C = 180; const long long M = (long long)H * W; const int ldq = 544;
  unsigned s = 7; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  std::vector<float> hq(M * ldq); for (auto& v : hq) v = rnd();
  std::vector<float> hr(6 * 1521); for (auto& v : hr) v = rnd();
  float *q, *out, *rel; unsigned long long* dbg;
  hipMalloc(&q, M * ldq * 4); hipMalloc(&out, M * 192 * 4); hipMalloc(&rel, hr.size() * 4);
  const size_t ndbg = 256 * 8 * 64; hipMalloc(&dbg, ndbg * 8);
  hipMemcpy(q, hq.data(), M * ldq * 4, hipMemcpyHostToDevice); hipMemcpy(rel, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
  auto run = [&]() { return ff_ocab_attn(q, ldq, 0, C, 2 * C, out, 192, 0, rel, 1, H, W, 6, 30, 16, 24, 0.1826f, 0, nullptr); };
  g_oc_dbg = nullptr;
  for (int i = 0; i < 3; ++i) if (run()) { printf("error: %s\n", g_err); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int i = 0; i < 20; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("ocab_attn: %.1f us per launch\n", ms * 1000 / 20);
  g_oc_dbg = dbg; hipMemset(dbg, 0, ndbg * 8); run(); hipDeviceSynchronize();
  std::vector<unsigned long long> t(ndbg); hipMemcpy(t.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull; for (size_t i = 0; i < ndbg; i += 64) if (t[i] && t[i] < t0) t0 = t[i];
  const char* names[64] = {}; names[0] = "start"; names[1] = "prologue (tables, stage 0)"; names[60] = "end";
  static char nb[64][40]; const char* ph[4] = {"next-stage loads issued", "9 tiles computed", "next stage stored", "barrier passed"};
  for (int n = 0; n < 4; ++n) for (int k = 0; k < 4; ++k) { snprintf(nb[2 + 4 * n + k], 40, "step %d %s", n, ph[k]); names[2 + 4 * n + k] = nb[2 + 4 * n + k]; }
  double prev = 0;
  for (int i = 0; i < 64; ++i) { if (!names[i]) continue; double sum = 0; int n = 0;
    for (size_t w8 = 0; w8 < 256 * 8; ++w8) { const unsigned long long v = t[w8 * 64 + i]; if (!v) continue; sum += (v - t0) / 100.0; ++n; }
    if (n) { printf("%-34s mean %7.2f us  +%6.2f\n", names[i], sum / n, sum / n - prev); prev = sum / n; } }
  return 0;
}

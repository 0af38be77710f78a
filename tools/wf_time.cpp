// Phase timing of win_attn_fused (debug aid, not part of the product): builds the kernel with -DWF_TIMING, runs the HAT
// geometry (256x256 tokens, 16x16 windows, 6 heads of 30) and prints wall-clock stamps per phase, averaged over waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWF_TIMING -Iimage-super-resolution-2_amd/csrc tools/wf_time.cpp -o tools/_dbg/wf_time
#include <stdarg.h>
#include <vector>
#include "../image-super-resolution-2_amd/csrc/win_attn_fused.hip"

static char g_err[512];
extern "C" const char* ff_last_error(void) { return g_err; }
void ff_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }

int main(int argc, char** argv) {
  const int nterms = argc > 1 ? atoi(argv[1]) : 1, shift = argc > 2 ? atoi(argv[2]) : 0, want_xn = argc > 3 ? atoi(argv[3]) : 1;
  const int H = 256, W = 256, C = 180, heads = 6, d = 30, ws = 16;
  const long long M = (long long)H * W;
  std::vector<float> hx(M * C), hrel(heads * 31 * 48), hb(3 * heads * 32, 0.f), hg(192, 1.f), hbe(192, 0.f);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hx) v = 2.f * rnd();
  for (auto& v : hrel) v = rnd();
  std::vector<unsigned short> hw((size_t)3 * heads * 2 * 32 * 192);
  for (auto& v : hw) { float f = 0.15f * rnd(); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  float *x, *out, *xn, *rel, *b, *g, *be; void* w; unsigned long long* dbg;
  hipMalloc(&x, M * C * 4); hipMalloc(&out, M * C * 4); hipMalloc(&xn, M * C * 4); hipMalloc(&rel, hrel.size() * 4);
  hipMalloc(&b, hb.size() * 4); hipMalloc(&g, 192 * 4); hipMalloc(&be, 192 * 4); hipMalloc(&w, hw.size() * 2);
  const size_t ndbg = 256 * 8 * 64;
  hipMalloc(&dbg, ndbg * 8);
  hipMemcpy(x, hx.data(), M * C * 4, hipMemcpyHostToDevice); hipMemcpy(rel, hrel.data(), hrel.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice); hipMemcpy(g, hg.data(), 192 * 4, hipMemcpyHostToDevice);
  hipMemcpy(be, hbe.data(), 192 * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  auto run = [&]() {
    return ff_win_attn_fused(x, C, out, C, 0, g, be, 1e-5f, w, b, rel, 31, 48, 1, H, W, H, W, ws, ws, shift, shift, shift > 0, 0, heads, d, C, 0,
                             want_xn ? xn : nullptr, C, nullptr, 0, 0, nterms, 0, 0, nullptr);
  };
  g_wf_dbg = nullptr;
  for (int i = 0; i < 3; ++i) if (run()) { printf("error: %s\n", g_err); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int i = 0; i < 20; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("nterms %d shift %d xn %d: %.1f us per launch (no stamps)\n", nterms, shift, want_xn, ms * 1000 / 20);
  g_wf_dbg = dbg; hipMemset(dbg, 0, ndbg * 8);
  run(); hipDeviceSynchronize();
  std::vector<unsigned long long> h(ndbg);
  hipMemcpy(h.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0;
  for (size_t i = 0; i < ndbg; i += 64) { if (h[i] && h[i] < t0) t0 = h[i]; if (h[i + 61] > t1) t1 = h[i + 61]; }
  printf("kernel span %.2f us (first start .. last drained), 100 MHz clock\n", (t1 - t0) / 100.0);
  const char* names[64] = {};
  names[0] = "start"; names[1] = "tables done"; names[2] = "x gathered + LN"; names[60] = "last stores issued"; names[61] = "stores drained";
  static char nb[64][48];
  const char* ph1[8] = {"q ring passed", "q gemm done", "k ring passed", "k gemm+store done", "v gemm+store done", "barrier A passed", "attention done", "barrier B passed"};
  const char* ph2[8] = {"barrier W passed", "q gemm done", "k gemm+store done", "v gemm+store done", "barrier A passed", "attention done", "vmcnt(0) passed", "epilogue issued"};
  const char* e = getenv("FF_WF_V2");
  const char** ph = (nterms == 1 && !(e && e[0] == '0')) ? ph2 : ph1;
  for (int hi = 0; hi < heads; ++hi) for (int k = 0; k < 8; ++k) { snprintf(nb[3 + 8 * hi + k], 48, "h%d %s", hi, ph[k]); names[3 + 8 * hi + k] = nb[3 + 8 * hi + k]; }
  double prev = 0;
  for (int i = 0; i < 64; ++i) {
    if (!names[i]) continue;
    double sum = 0, mn = 1e30, mx = 0; int n = 0;
    for (size_t w8 = 0; w8 < 256 * 8; ++w8) { const unsigned long long v = h[w8 * 64 + i]; if (!v) continue; const double u = (v - t0) / 100.0; sum += u; if (u < mn) mn = u; if (u > mx) mx = u; ++n; }
    if (!n) continue;
    printf("%-26s mean %7.2f us  (min %7.2f max %7.2f)  +%6.2f\n", names[i], sum / n, mn, mx, sum / n - prev);
    prev = sum / n;
  }
  return 0;
}

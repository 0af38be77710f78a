// 3x3 / stride 1 / pad 1 convolution for SMALL channel counts (Cin <= 64, Cout <= 16) on the fp32 VALU:
//     out = res + alpha * act( conv3x3(in) + bias )
// The fusion stack ends its branches with such layers at 1024 x 1024 (to_rgb 32->16->3 hierarchical_fusion.py:124-128, refine_net
// 64->3 enhanced_fusion.py:288, edge fusion 32->3 / gates 6->16->1 / 8->1 edge_enhancement.py:112,170-180).  On the MFMA
// implicit-GEMM path a 32-wide N tile is 80-97 % padding and the scalar gather of Cin = 3 / 6 / 16 rows runs at 0.3-0.5 TB/s
// (160-205 us per layer); here a 16x16-pixel workgroup stages its 18x18 halo tile in LDS once (pixel pitch Cin + 4 floats:
// conflict-free 16-byte reads), a thread owns one pixel and all output channels, and the weights are wave-uniform scalar
// loads ([tap][ci][CT] fp32), so the inner loop is one LDS read per 4 input channels and CT FMAs per input value.
// Exact fp32 accumulation (no bf16 split needed: there is no MFMA to feed).
#include "ff_common.h"

struct SmallConvParams {
  const float* in; const float* w; const float* bias; const float* res; float* out;
  int B, H, W, Cin, CinP, ldi, Cout, ldo, ldr, act;
  float alpha;
};

template <int CT>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(SmallConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float tile[];         // [18][18][CinP + 4]
  const int PS = p.CinP + 4;
  const int tid = threadIdx.x;
  const int tx = blockIdx.x, ty = blockIdx.y, b = blockIdx.z;
  const int x0 = tx * 16 - 1, y0 = ty * 16 - 1;
  const int c4n = p.CinP >> 2;
  const bool vec = (p.Cin & 3) == 0 && (p.ldi & 3) == 0 && ((((uintptr_t)p.in) & 15) == 0);
  for (int i = tid; i < 324 * c4n; i += 256) {
    const int pix = i / c4n, c = (i - pix * c4n) * 4;
    const int iy = y0 + pix / 18, ix = x0 + pix % 18;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
      const float* src = p.in + ((long long)(b * p.H + iy) * p.W + ix) * p.ldi + c;
      if (vec) v = *reinterpret_cast<const f32x4*>(src);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (c + e < p.Cin) ? src[e] : 0.f;
      }
    }
    *reinterpret_cast<f32x4*>(tile + pix * PS + c) = v;
  }
  __syncthreads();
  const int px = tid & 15, py = tid >> 4;
  float acc[CT];
#pragma unroll
  for (int o = 0; o < CT; ++o) acc[o] = 0.f;
  const float* __restrict__ w = p.w;
  for (int tap = 0; tap < 9; ++tap) {
    const float* ip = tile + ((py + tap / 3) * 18 + px + tap % 3) * PS;
    const float* wt = w + (long long)tap * p.CinP * CT;
    for (int c4 = 0; c4 < c4n; ++c4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(ip + 4 * c4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = 0; o < CT; ++o) acc[o] = __builtin_fmaf(v[e], wt[(4 * c4 + e) * CT + o], acc[o]);     // wt[...]: wave-uniform -> scalar loads
    }
  }
  const int ox = tx * 16 + px, oy = ty * 16 + py;
  if (ox < p.W && oy < p.H) {
    const long long pix = (long long)(b * p.H + oy) * p.W + ox;
#pragma unroll
    for (int o = 0; o < CT; ++o)
      if (o < p.Cout) {
        const float v = ff_act(acc[o] + (p.bias ? p.bias[o] : 0.f), p.act) * p.alpha;
        p.out[pix * p.ldo + o] = (p.res ? p.res[pix * p.ldr + o] : 0.f) + v;
      }
  }
}

// w_small: fp32 [9][CinP][CT] with CinP = ceil4(Cin), CT = 1 / 4 / 16 >= Cout (zero padded): prep.pack_conv3x3_small
extern "C" int ff_conv3x3_small(const float* in, int ldi, const float* w_small, int ct, const float* bias, const float* res, int ldr,
                                float* out, int ldo, int B, int H, int W, int Cin, int Cout, int act, float alpha, void* stream) {
  FF_CHECK_ARG(in && w_small && out && in != out, "ff_conv3x3_small: null pointer / in-place");
  FF_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 64 && Cout > 0 && Cout <= ct && (ct == 1 || ct == 4 || ct == 16), "ff_conv3x3_small: needs Cin <= 64, Cout <= CT in {1, 4, 16}");
  FF_CHECK_ARG(ldi >= Cin && ldo >= Cout && (!res || ldr >= Cout), "ff_conv3x3_small: row pitches too small");
  FF_CHECK_ARG((H + 15) / 16 < 65536 && B < 65536, "ff_conv3x3_small: image too large");
  SmallConvParams p;
  p.in = in; p.w = w_small; p.bias = bias; p.res = res; p.out = out; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.CinP = (Cin + 3) / 4 * 4;
  p.ldi = ldi; p.Cout = Cout; p.ldo = ldo; p.ldr = ldr; p.act = act; p.alpha = alpha;
  const dim3 grid((W + 15) / 16, (H + 15) / 16, B);
  const size_t lds = (size_t)324 * (p.CinP + 4) * 4;
#define SC_LAUNCH(CTV)                                                                                                      \
  do {                                                                                                                      \
    static bool attr_set = false;                                                                                           \
    if (!attr_set) {                                                                                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_small_kernel<CTV>),                         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 324 * 68 * 4);                         \
      if (e != hipSuccess) { ff_set_error("ff_conv3x3_small: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; } \
      attr_set = true;                                                                                                      \
    }                                                                                                                       \
    hipLaunchKernelGGL(conv3x3_small_kernel<CTV>, grid, dim3(256), lds, (hipStream_t)stream, p);                            \
  } while (0)
  if (ct == 1) SC_LAUNCH(1); else if (ct == 4) SC_LAUNCH(4); else SC_LAUNCH(16);
#undef SC_LAUNCH
  FF_LAUNCH_CHECK("ff_conv3x3_small");
  return FF_OK;
}

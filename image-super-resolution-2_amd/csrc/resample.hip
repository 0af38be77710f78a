// Resamplers with PyTorch's exact index/weight conventions (ATen UpSample.h):
//   bilinear, align_corners=False:  src = max(scale*(dst+0.5)-0.5, 0); i0 = min(floor(src), n-1); i1 = min(i0+1, n-1)
//   bicubic,  align_corners=False:  src = scale*(dst+0.5)-0.5 (not clamped), A = -0.75, border-clamped taps
//   scale = 1/scale_factor when the caller passed scale_factor (nafnet/__init__.py:128-133 x4 bicubic,
//   fusion_network.py:590,595 x0.5 / x0.25), else in/out (every F.interpolate(size=...) site:
//   hierarchical_fusion.py:147-181, enhanced_fusion.py:550,620,640,675, edge_enhancement.py:203,245,
//   multi_domain_frequency.py:296,366).  The host computes `scale` exactly as ATen does and passes it.
// Generic element strides on both sides let one kernel serve planar (NCHW) and NHWC tensors and write
// straight into a channel slice of a wider NHWC tensor.  HBM-bound.
#include "ff_common.h"

struct ResizeParams {
  const float* in; float* out;
  long long isb, isc, isy, isx, osb, osc, osy, osx;
  int B, C, Hi, Wi, Ho, Wo;
  float sh, sw;
  int mode;
  float mul;
};

__device__ __forceinline__ void cubic_coeffs(float t, float w[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x3 = 2.f - t, u = 1.f - t;
  w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
  w[2] = ((A + 2.f) * u - (A + 3.f)) * u * u + 1.f;
  w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ __launch_bounds__(256) void resize_kernel(ResizeParams p) {
  const long long total = (long long)p.B * p.Ho * p.Wo * p.C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    int c, x, y, b;
    if (p.osc == 1) {        // channel fastest (NHWC out)
      c = (int)(i % p.C); long long t = i / p.C;
      x = (int)(t % p.Wo); t /= p.Wo;
      y = (int)(t % p.Ho); b = (int)(t / p.Ho);
    } else {                 // x fastest (planar out)
      x = (int)(i % p.Wo); long long t = i / p.Wo;
      y = (int)(t % p.Ho); t /= p.Ho;
      c = (int)(t % p.C); b = (int)(t / p.C);
    }
    const float* src = p.in + b * p.isb + c * p.isc;
    float v;
    if (p.mode == 0) {
      int y0, y1, x0, x1; float ly, lx;
      if (p.Ho == p.Hi) { y0 = y1 = y; ly = 0.f; }
      else {
        float sy = fmaxf(p.sh * ((float)y + 0.5f) - 0.5f, 0.f);
        y0 = min((int)floorf(sy), p.Hi - 1); y1 = min(y0 + 1, p.Hi - 1);
        ly = fminf(fmaxf(sy - (float)y0, 0.f), 1.f);
      }
      if (p.Wo == p.Wi) { x0 = x1 = x; lx = 0.f; }
      else {
        float sx = fmaxf(p.sw * ((float)x + 0.5f) - 0.5f, 0.f);
        x0 = min((int)floorf(sx), p.Wi - 1); x1 = min(x0 + 1, p.Wi - 1);
        lx = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
      }
      const float a = src[y0 * p.isy + x0 * p.isx], bq = src[y0 * p.isy + x1 * p.isx];
      const float cq = src[y1 * p.isy + x0 * p.isx], d = src[y1 * p.isy + x1 * p.isx];
      v = (1.f - ly) * ((1.f - lx) * a + lx * bq) + ly * ((1.f - lx) * cq + lx * d);
    } else {
      const float sy = p.sh * ((float)y + 0.5f) - 0.5f, sx = p.sw * ((float)x + 0.5f) - 0.5f;
      const int iy = (int)floorf(sy), ix = (int)floorf(sx);
      float wy[4], wx[4];
      cubic_coeffs(sy - (float)iy, wy);
      cubic_coeffs(sx - (float)ix, wx);
      v = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int yy = min(max(iy - 1 + j, 0), p.Hi - 1);
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int xx = min(max(ix - 1 + k, 0), p.Wi - 1);
          r += wx[k] * src[yy * p.isy + xx * p.isx];
        }
        v += wy[j] * r;
      }
    }
    p.out[b * p.osb + c * p.osc + y * p.osy + x * p.osx] = v * p.mul;
  }
}

// Bilinear, channel-last on both sides, four channels per thread: the index / weight arithmetic is shared by a float4 and every
// access is 16 bytes (the generic kernel above moves one float per thread: ~1 TB/s on the 64-channel up-samplings of the
// hierarchical fusion, hierarchical_fusion.py:147-181).
__global__ __launch_bounds__(256) void resize_bilinear_nhwc4_kernel(ResizeParams p) {
  const int c4n = p.C >> 2;
  const long long total = (long long)p.B * p.Ho * p.Wo * c4n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4; long long t = i / c4n;
    const int x = (int)(t % p.Wo); t /= p.Wo;
    const int y = (int)(t % p.Ho); const int b = (int)(t / p.Ho);
    int y0, y1, x0, x1; float ly, lx;
    if (p.Ho == p.Hi) { y0 = y1 = y; ly = 0.f; }
    else {
      const float sy = fmaxf(p.sh * ((float)y + 0.5f) - 0.5f, 0.f);
      y0 = min((int)floorf(sy), p.Hi - 1); y1 = min(y0 + 1, p.Hi - 1);
      ly = fminf(fmaxf(sy - (float)y0, 0.f), 1.f);
    }
    if (p.Wo == p.Wi) { x0 = x1 = x; lx = 0.f; }
    else {
      const float sx = fmaxf(p.sw * ((float)x + 0.5f) - 0.5f, 0.f);
      x0 = min((int)floorf(sx), p.Wi - 1); x1 = min(x0 + 1, p.Wi - 1);
      lx = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
    }
    const float* src = p.in + b * p.isb + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + y0 * p.isy + x0 * p.isx), bq = *reinterpret_cast<const f32x4*>(src + y0 * p.isy + x1 * p.isx);
    const f32x4 cq = *reinterpret_cast<const f32x4*>(src + y1 * p.isy + x0 * p.isx), d = *reinterpret_cast<const f32x4*>(src + y1 * p.isy + x1 * p.isx);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e)            // same association as the scalar kernel: bit-identical results
      v[e] = ((1.f - ly) * ((1.f - lx) * a[e] + lx * bq[e]) + ly * ((1.f - lx) * cq[e] + lx * d[e])) * p.mul;
    *reinterpret_cast<f32x4*>(p.out + b * p.osb + c + y * p.osy + x * p.osx) = v;
  }
}

extern "C" int ff_resize(const float* in, long long isb, long long isc, long long isy, long long isx, int Hi, int Wi,
                         float* out, long long osb, long long osc, long long osy, long long osx, int Ho, int Wo, int B,
                         int C, float scale_h, float scale_w, int mode, float mul, void* stream) {
  FF_CHECK_ARG(in && out && B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "ff_resize: bad args");
  FF_CHECK_ARG(mode == 0 || mode == 1, "ff_resize: mode must be 0 (bilinear) or 1 (bicubic)");
  ResizeParams p;
  p.in = in; p.out = out; p.isb = isb; p.isc = isc; p.isy = isy; p.isx = isx;
  p.osb = osb; p.osc = osc; p.osy = osy; p.osx = osx;
  p.B = B; p.C = C; p.Hi = Hi; p.Wi = Wi; p.Ho = Ho; p.Wo = Wo; p.sh = scale_h; p.sw = scale_w; p.mode = mode; p.mul = mul;
  const bool v4 = mode == 0 && isc == 1 && osc == 1 && C % 4 == 0 && isx % 4 == 0 && isy % 4 == 0 && isb % 4 == 0 && osx % 4 == 0 && osy % 4 == 0 &&
                  osb % 4 == 0 && (((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & 15) == 0;
  long long nb = ((long long)B * Ho * Wo * (v4 ? C / 4 : C) + 255) / 256;
  if (nb > 16384) nb = 16384;
  if (v4) hipLaunchKernelGGL(resize_bilinear_nhwc4_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(resize_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
  FF_LAUNCH_CHECK("ff_resize");
  return FF_OK;
}

// 2x2 average pool, stride 2, NHWC (edge_enhancement.py:202)
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo,
                                                       int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const long long total = (long long)B * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho); const int b = (int)(t / Ho);
    const float* s = in + (((long long)b * H + 2 * y) * W + 2 * x) * ldi + c;
    out[(((long long)b * Ho + y) * Wo + x) * ldo + c] = (s[0] + s[ldi] + s[(long long)W * ldi] + s[(long long)(W + 1) * ldi]) * 0.25f;
  }
}

extern "C" int ff_avgpool2(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, void* stream) {
  FF_CHECK_ARG(in && out && B > 0 && H >= 2 && W >= 2 && C > 0 && ldi >= C && ldo >= C, "ff_avgpool2: bad args");
  long long nb = ((long long)B * (H / 2) * (W / 2) * C + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, ldi, out, ldo, B, H, W, C);
  FF_LAUNCH_CHECK("ff_avgpool2");
  return FF_OK;
}

// Depth-wise convolutions on NHWC tensors (HBM-bound: one read of the tile + halo, one write):
//   3x3 (+BN+GELU)  dat_arch.py:403-407,109 ; nafnet_arch.py:78-81
//   5x5, 1x21, 21x1 large_kernel_attention.py:59-78 (LKA chain)
//   Gaussian 5x5    edge_enhancement.py:62
// Channels are the fastest axis, so a wave reads 64 consecutive channels (or 16 float4) of one tap
// coalesced; the kh*kw taps of neighbouring pixels are served from L1/L2.  Weights are stored
// tap-major [kh*kw][C] so the per-tap weight vector is also one coalesced load.
//   out = act( (sum_taps w*x + bias) * post_scale + post_shift ) * mul_in      (mul_in optional: the SGFN gate, dat_arch.py:123)
#include "ff_common.h"

struct DwParams {
  const float* in; float* out; const float* w; const float* bias; const float* ps; const float* pt; const float* mulin;
  int ldm, ldi, ldo, B, H, W, C, Ho, Wo, KH, KW, sy, sx, py, px, act;
};

__global__ __launch_bounds__(256) void dwconv_vec4_kernel(DwParams p) {
  const int c4n = p.C >> 2;
  const long long total = (long long)p.B * p.Ho * p.Wo * c4n;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % c4n) * 4;
    long long pix = idx / c4n;
    const int ox = (int)(pix % p.Wo); pix /= p.Wo;
    const int oy = (int)(pix % p.Ho);
    const int b = (int)(pix / p.Ho);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < p.KH; ++ky) {
      const int iy = oy * p.sy - p.py + ky;
      if ((unsigned)iy >= (unsigned)p.H) continue;
      for (int kx = 0; kx < p.KW; ++kx) {
        const int ix = ox * p.sx - p.px + kx;
        if ((unsigned)ix >= (unsigned)p.W) continue;
        const f32x4 x = *reinterpret_cast<const f32x4*>(p.in + ((long long)(b * p.H + iy) * p.W + ix) * p.ldi + c);
        const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + (long long)(ky * p.KW + kx) * p.C + c);
        acc += x * w;
      }
    }
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[e] + (p.bias ? p.bias[c + e] : 0.f);
      if (p.ps) v = v * p.ps[c + e] + p.pt[c + e];
      r[e] = ff_act(v, p.act);
    }
    if (p.mulin) r *= *reinterpret_cast<const f32x4*>(p.mulin + ((long long)(b * p.Ho + oy) * p.Wo + ox) * p.ldm + c);
    *reinterpret_cast<f32x4*>(p.out + ((long long)(b * p.Ho + oy) * p.Wo + ox) * p.ldo + c) = r;
  }
}

// 3x3 / stride 1 / pad 1 specialisation (DAT's two depth-wise convs, dat_arch.py:109,403): a thread owns 4 channels of a
// 4-row output strip, keeps the 9 tap weights in registers and slides a 3-column window down 6 input rows, so each
// output costs 4.5 float4 loads instead of 9 and no per-tap weight loads or bounds branches.
template <int ACT>
__global__ __launch_bounds__(256, 4) void dwconv3x3_strip_kernel(DwParams p) {
  const int c4n = p.C >> 2;
  const int nstrip = (p.H + 3) >> 2;
  const long long total = (long long)p.B * nstrip * p.W * c4n;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % c4n) * 4;
    long long t = idx / c4n;
    const int ox = (int)(t % p.W); t /= p.W;
    const int y0 = (int)(t % nstrip) * 4;
    const int b = (int)(t / nstrip);
    f32x4 w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const f32x4*>(p.w + (long long)k * p.C + c);
    f32x4 bias = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c) : z4;
    f32x4 ps = {1.f, 1.f, 1.f, 1.f}, pt = z4;
    if (p.ps) { ps = *reinterpret_cast<const f32x4*>(p.ps + c); pt = *reinterpret_cast<const f32x4*>(p.pt + c); }
    // one input row at a time feeding up to three output rows (see dwconv3x3_ln_strip_kernel): same (ky, kx) accumulation order per
    // output, a third fewer live registers, four waves per SIMD
    f32x4 acc4[4] = {z4, z4, z4, z4};
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int iy = y0 - 1 + r;
      const bool oky = (unsigned)iy < (unsigned)p.H;
      f32x4 t[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int ix = ox - 1 + k;
        const bool ok = oky && (unsigned)ix < (unsigned)p.W;
        const f32x4 u = *reinterpret_cast<const f32x4*>(p.in + (ok ? ((long long)(b * p.H + iy) * p.W + ix) * p.ldi + c : 0));
        t[k] = ok ? u : z4;
      }
#pragma unroll
      for (int ky = 2; ky >= 0; --ky) {
        const int o = r - ky;
        if (o >= 0 && o < 4) {
          acc4[o] += t[0] * w[ky * 3];
          acc4[o] += t[1] * w[ky * 3 + 1];
          acc4[o] += t[2] * w[ky * 3 + 2];
        }
      }
      asm volatile("" : "+v"(acc4[0]), "+v"(acc4[1]), "+v"(acc4[2]), "+v"(acc4[3]) : : "memory");
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (y0 + r >= p.H) break;
      const long long pix = (long long)(b * p.H + y0 + r) * p.W + ox;
      const f32x4 acc = (acc4[r] + bias) * ps + pt;
      const f32x4 m = p.mulin ? *reinterpret_cast<const f32x4*>(p.mulin + pix * p.ldm + c) : (f32x4){1.f, 1.f, 1.f, 1.f};
      f32x4 o;
#pragma unroll
      // GELU through the 1.5e-7-accurate erf of ff_gelu_fast: the libm erff costs ~55 VALU ops with divergent branches and made
      // this HBM-bound kernel VALU-bound (66 us for 94 MB at 65 536 x 180; r2)
      for (int e = 0; e < 4; ++e) o[e] = ff_act_c<ACT, true>(acc[e]) * m[e];
      *reinterpret_cast<f32x4*>(p.out + pix * p.ldo + c) = o;
    }
  }
}

// Register-tiled depth-wise convolution, stride 1, "same" padding, for the large kernels of the LKA chain
// (large_kernel_attention.py:59-78: 5x5, 1x21, 21x1 on 9 x 64 channels) and the 5x5 Gaussian (edge_enhancement.py:62):
// a thread owns 4 channels of an RY x RX output patch and loads its (RY+KH-1) x (RX+KW-1) input window ONCE -- 3.5 (1x21), 3.5 (21x1)
// and 9 (5x5) float4 loads per output instead of one per tap (the generic kernel ran the chain at 1 TB/s).
template <int KH, int KW, int RY, int RX>
__global__ __launch_bounds__(256) void dwconv_tile_kernel(DwParams p) {
  constexpr int IY = RY + KH - 1, IX = RX + KW - 1;
  const int c4n = p.C >> 2;
  const int ny = (p.H + RY - 1) / RY, nx = (p.W + RX - 1) / RX;
  const long long total = (long long)p.B * ny * nx * c4n;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % c4n) * 4;
    long long t = idx / c4n;
    const int x0 = (int)(t % nx) * RX; t /= nx;
    const int y0 = (int)(t % ny) * RY;
    const int b = (int)(t / ny);
    f32x4 in[IY][IX];
#pragma unroll
    for (int r = 0; r < IY; ++r) {
      const int iy = y0 - p.py + r;
      const bool oky = (unsigned)iy < (unsigned)p.H;
#pragma unroll
      for (int k = 0; k < IX; ++k) {
        const int ix = x0 - p.px + k;
        const bool ok = oky && (unsigned)ix < (unsigned)p.W;
        const f32x4 u = *reinterpret_cast<const f32x4*>(p.in + (ok ? ((long long)(b * p.H + iy) * p.W + ix) * p.ldi + c : 0));
        in[r][k] = ok ? u : z4;
      }
    }
    f32x4 acc[RY][RX];
#pragma unroll
    for (int r = 0; r < RY; ++r)
#pragma unroll
      for (int k = 0; k < RX; ++k) acc[r][k] = z4;
#pragma unroll
    for (int ky = 0; ky < KH; ++ky)
#pragma unroll
      for (int kx = 0; kx < KW; ++kx) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + (long long)(ky * KW + kx) * p.C + c);
#pragma unroll
        for (int r = 0; r < RY; ++r)
#pragma unroll
          for (int k = 0; k < RX; ++k) acc[r][k] += in[r + ky][k + kx] * w;
      }
    const f32x4 bias = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c) : z4;
    f32x4 ps = {1.f, 1.f, 1.f, 1.f}, pt = z4;
    if (p.ps) { ps = *reinterpret_cast<const f32x4*>(p.ps + c); pt = *reinterpret_cast<const f32x4*>(p.pt + c); }
#pragma unroll
    for (int r = 0; r < RY; ++r)
#pragma unroll
      for (int k = 0; k < RX; ++k) {
        if (y0 + r >= p.H || x0 + k >= p.W) continue;
        const long long pix = (long long)(b * p.H + y0 + r) * p.W + x0 + k;
        f32x4 v = (acc[r][k] + bias) * ps + pt;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ff_act(v[e], p.act);
        if (p.mulin) v *= *reinterpret_cast<const f32x4*>(p.mulin + pix * p.ldm + c);
        *reinterpret_cast<f32x4*>(p.out + pix * p.ldo + c) = v;
      }
  }
}

// The same strip kernel with LayerNorm applied ON LOAD (DAT SpatialGate, dat_arch.py:117-122: x2 -> LayerNorm -> dw3x3, times x1):
// the per-token (mean, rstd) come from the producer's epilogue (ff_token_linear stats_out), gamma / beta per channel; the
// convolution's zero padding applies to the NORMALISED tensor, so out-of-image taps contribute exactly zero.
struct DwLnParams {
  const float* in; float* out; const float* w; const float* bias; const float* stats; const float* gamma; const float* beta;
  const float* mulin;
  int ldm, ldi, ldo, B, H, W, C;
};

__global__ __launch_bounds__(256, 4) void dwconv3x3_ln_strip_kernel(DwLnParams p) {
  const int c4n = p.C >> 2;
  const int nstrip = (p.H + 3) >> 2;
  const long long total = (long long)p.B * nstrip * p.W * c4n;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % c4n) * 4;
    long long t = idx / c4n;
    const int ox = (int)(t % p.W); t /= p.W;
    const int y0 = (int)(t % nstrip) * 4;
    const int b = (int)(t / nstrip);
    f32x4 w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const f32x4*>(p.w + (long long)k * p.C + c);
    const f32x4 bias = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c) : z4;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gamma + c), b4 = *reinterpret_cast<const f32x4*>(p.beta + c);
    // One input row at a time (three normalised elements) feeding up to three output rows: the whole 6 x 3 window never sits in
    // registers next to the nine taps (136 VGPRs -> under 128 = four waves per SIMD for this bandwidth-bound kernel); each output
    // still accumulates its taps in (ky, kx) order, so the result is bit-identical to the windowed form.
    f32x4 acc4[4] = {z4, z4, z4, z4};
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int iy = y0 - 1 + r;
      const bool oky = (unsigned)iy < (unsigned)p.H;
      f32x4 t[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int ix = ox - 1 + k;
        const bool ok = oky && (unsigned)ix < (unsigned)p.W;
        const long long tk = ok ? (long long)(b * p.H + iy) * p.W + ix : 0;
        const f32x4 u = *reinterpret_cast<const f32x4*>(p.in + tk * p.ldi + c);
        const float2 ms = *reinterpret_cast<const float2*>(p.stats + 2 * tk);
        t[k] = ok ? (u - ms.x) * ms.y * g4 + b4 : z4;
      }
#pragma unroll
      for (int ky = 2; ky >= 0; --ky) {                          // output row o = r - ky receives its tap row ky
        const int o = r - ky;
        if (o >= 0 && o < 4) {
          acc4[o] += t[0] * w[ky * 3];
          acc4[o] += t[1] * w[ky * 3 + 1];
          acc4[o] += t[2] * w[ky * 3 + 2];
        }
      }
      asm volatile("" : "+v"(acc4[0]), "+v"(acc4[1]), "+v"(acc4[2]), "+v"(acc4[3]) : : "memory");
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc4[r] += bias;
    // the gate operand is fetched only now (window and taps are dead): 128 VGPRs = four waves per SIMD for this bandwidth-bound kernel
    asm volatile("" : "+v"(acc4[0]), "+v"(acc4[1]), "+v"(acc4[2]), "+v"(acc4[3]) : : "memory");
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (y0 + r >= p.H) break;
      const long long pix = (long long)(b * p.H + y0 + r) * p.W + ox;
      const f32x4 m = p.mulin ? *reinterpret_cast<const f32x4*>(p.mulin + pix * p.ldm + c) : (f32x4){1.f, 1.f, 1.f, 1.f};
      *reinterpret_cast<f32x4*>(p.out + pix * p.ldo + c) = acc4[r] * m;
    }
  }
}

extern "C" int ff_dwconv3x3_ln(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, const float* w_tapmajor,
                               const float* bias, const float* stats, const float* gamma, const float* beta, const float* mul_in,
                               int ldm, void* stream) {
  FF_CHECK_ARG(in && out && w_tapmajor && stats && gamma && beta && in != out, "ff_dwconv3x3_ln: null pointer / in-place");
  FF_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ldi >= C && ldo >= C && ldi % 4 == 0 && ldo % 4 == 0, "ff_dwconv3x3_ln: bad dims");
  FF_CHECK_ARG((((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & 15) == 0 && (((uintptr_t)w_tapmajor) & 15) == 0 && (((uintptr_t)gamma) & 15) == 0 &&
               (((uintptr_t)beta) & 15) == 0 && (((uintptr_t)stats) & 7) == 0 && (!bias || (((uintptr_t)bias) & 15) == 0), "ff_dwconv3x3_ln: alignment");
  FF_CHECK_ARG(!mul_in || (ldm >= C && ldm % 4 == 0 && (((uintptr_t)mul_in) & 15) == 0), "ff_dwconv3x3_ln: mul_in rows must be 16-byte aligned");
  DwLnParams p;
  p.in = in; p.out = out; p.w = w_tapmajor; p.bias = bias; p.stats = stats; p.gamma = gamma; p.beta = beta; p.mulin = mul_in;
  p.ldm = ldm; p.ldi = ldi; p.ldo = ldo; p.B = B; p.H = H; p.W = W; p.C = C;
  const long long tot = (long long)B * ((H + 3) / 4) * W * (C / 4);
  long long nbs = (tot + 255) / 256;
  if (nbs > 256 * 32) nbs = 256 * 32;
  hipLaunchKernelGGL(dwconv3x3_ln_strip_kernel, dim3((unsigned)nbs), dim3(256), 0, (hipStream_t)stream, p);
  FF_LAUNCH_CHECK("ff_dwconv3x3_ln");
  return FF_OK;
}

__global__ __launch_bounds__(256) void dwconv_scalar_kernel(DwParams p) {
  const long long total = (long long)p.B * p.Ho * p.Wo * p.C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % p.C);
    long long pix = idx / p.C;
    const int ox = (int)(pix % p.Wo); pix /= p.Wo;
    const int oy = (int)(pix % p.Ho);
    const int b = (int)(pix / p.Ho);
    float acc = 0.f;
    for (int ky = 0; ky < p.KH; ++ky) {
      const int iy = oy * p.sy - p.py + ky;
      if ((unsigned)iy >= (unsigned)p.H) continue;
      for (int kx = 0; kx < p.KW; ++kx) {
        const int ix = ox * p.sx - p.px + kx;
        if ((unsigned)ix >= (unsigned)p.W) continue;
        acc += p.in[((long long)(b * p.H + iy) * p.W + ix) * p.ldi + c] * p.w[(long long)(ky * p.KW + kx) * p.C + c];
      }
    }
    float v = acc + (p.bias ? p.bias[c] : 0.f);
    if (p.ps) v = v * p.ps[c] + p.pt[c];
    float rr = ff_act(v, p.act);
    if (p.mulin) rr *= p.mulin[((long long)(b * p.Ho + oy) * p.Wo + ox) * p.ldm + c];
    p.out[((long long)(b * p.Ho + oy) * p.Wo + ox) * p.ldo + c] = rr;
  }
}

extern "C" int ff_dwconv2d(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, int Ho, int Wo,
                           const float* w_tapmajor, const float* bias, int KH, int KW, int sy, int sx, int py, int px,
                           const float* post_scale, const float* post_shift, int act, const float* mul_in, int ldm, void* stream) {
  FF_CHECK_ARG(in && out && w_tapmajor, "ff_dwconv2d: null pointer");
  FF_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && ldi >= C && ldo >= C, "ff_dwconv2d: bad dims");
  FF_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "ff_dwconv2d: post scale/shift must come together");
  DwParams p;
  p.in = in; p.out = out; p.w = w_tapmajor; p.bias = bias; p.ps = post_scale; p.pt = post_shift; p.mulin = mul_in; p.ldm = ldm;
  p.ldi = ldi; p.ldo = ldo; p.B = B; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo;
  p.KH = KH; p.KW = KW; p.sy = sy; p.sx = sx; p.py = py; p.px = px; p.act = act;
  FF_CHECK_ARG(!mul_in || ldm >= C, "ff_dwconv2d: ldm too small");
  const bool v4 = (C % 4 == 0) && (ldi % 4 == 0) && (ldo % 4 == 0) && (!mul_in || (ldm % 4 == 0 && ((uintptr_t)mul_in & 15) == 0)) && (((uintptr_t)in & 15) == 0) &&
                  (((uintptr_t)out & 15) == 0) && (((uintptr_t)w_tapmajor & 15) == 0);
  if (v4 && KH == 3 && KW == 3 && sy == 1 && sx == 1 && py == 1 && px == 1 && Ho == H && Wo == W && out != in) {
    const long long tot = (long long)B * ((H + 3) / 4) * W * (C / 4);
    long long nbs = (tot + 255) / 256;
    if (nbs > 256 * 32) nbs = 256 * 32;
    auto go = [&](auto A) {
      hipLaunchKernelGGL((dwconv3x3_strip_kernel<decltype(A)::value>), dim3((unsigned)nbs), dim3(256), 0, (hipStream_t)stream, p);
    };
    FF_DISPATCH_ACT(act, go);
    FF_LAUNCH_CHECK("ff_dwconv2d");
    return FF_OK;
  }
  if (v4 && sy == 1 && sx == 1 && Ho == H && Wo == W && out != in && py == KH / 2 && px == KW / 2 &&
      ((KH == 1 && KW == 21) || (KH == 21 && KW == 1) || (KH == 5 && KW == 5))) {
    const int ry = KH == 21 ? 8 : (KH == 5 ? 2 : 1), rx = KW == 21 ? 8 : (KW == 5 ? 2 : 1);
    const long long tot = (long long)B * ((H + ry - 1) / ry) * ((W + rx - 1) / rx) * (C / 4);
    long long nbt = (tot + 255) / 256;
    if (nbt > 256 * 32) nbt = 256 * 32;
    if (KW == 21) hipLaunchKernelGGL((dwconv_tile_kernel<1, 21, 1, 8>), dim3((unsigned)nbt), dim3(256), 0, (hipStream_t)stream, p);
    else if (KH == 21) hipLaunchKernelGGL((dwconv_tile_kernel<21, 1, 8, 1>), dim3((unsigned)nbt), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((dwconv_tile_kernel<5, 5, 2, 2>), dim3((unsigned)nbt), dim3(256), 0, (hipStream_t)stream, p);
    FF_LAUNCH_CHECK("ff_dwconv2d");
    return FF_OK;
  }
  const long long total = (long long)B * Ho * Wo * (v4 ? C / 4 : C);
  long long nb = (total + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  if (v4)
    hipLaunchKernelGGL(dwconv_vec4_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(dwconv_scalar_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
  FF_LAUNCH_CHECK("ff_dwconv2d");
  return FF_OK;
}

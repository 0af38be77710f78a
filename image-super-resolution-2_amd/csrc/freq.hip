// Frequency-band decomposition of the LR image (multi_domain_frequency.py:146-196 DCT, :273-299 DWT,
// :352-385 FFT).  Input is the planar [C][H][W] LR image (B folded into C); the nine bands are written
// into one NHWC tensor [H][W][27] (band-major channel groups) so the consumers (band_proj 1x1,
// band fusion 27-channel concat) address slices in place.  All HBM/LDS-bound, tiny FLOP counts.
#include "ff_common.h"

__device__ __forceinline__ int reflect_i(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// ------------------------------------------------------------------------------------------ DCT
// One 8x8 block per 64 threads, 4 blocks per workgroup.  coef = D (X D^T); band_b = D^T ((coef.mask_b) D).
__global__ __launch_bounds__(256) void dct8_bands_kernel(const float* __restrict__ x, int C, int H, int W,
                                                         const float* __restrict__ D, const float* __restrict__ masks,
                                                         const float* __restrict__ band_scale, float* __restrict__ out,
                                                         int ldo, int ch_off) {
  __shared__ float sD[64], sX[4][64], sT[4][64], sC[4][64];
  const int tid = threadIdx.x, g = tid >> 6, e = tid & 63, i = e >> 3, j = e & 7;
  const int nbx = (W + 7) / 8, nby = (H + 7) / 8;
  const long long blk = (long long)blockIdx.x * 4 + g;
  const long long nblk = (long long)C * nby * nbx;
  if (tid < 64) sD[tid] = D[tid];
  const bool live = blk < nblk;
  int c = 0, by = 0, bx = 0;
  if (live) { bx = (int)(blk % nbx); long long t = blk / nbx; by = (int)(t % nby); c = (int)(t / nby); }
  const int yy = by * 8 + i, xx = bx * 8 + j;
  sX[g][e] = live ? x[((long long)c * H + reflect_i(yy, H)) * W + reflect_i(xx, W)] : 0.f;
  __syncthreads();
  float s = 0.f;                                            // T = X D^T : T[i][v] = sum_j X[i][j] D[v][j]
#pragma unroll
  for (int k = 0; k < 8; ++k) s += sX[g][i * 8 + k] * sD[j * 8 + k];
  sT[g][e] = s;
  __syncthreads();
  s = 0.f;                                                  // coef[u][v] = sum_i D[u][i] T[i][v]
#pragma unroll
  for (int k = 0; k < 8; ++k) s += sD[i * 8 + k] * sT[g][k * 8 + j];
  sC[g][e] = s;
  __syncthreads();
  for (int b = 0; b < 3; ++b) {
    // U = (coef.mask) D : U[u][j] = sum_v cm[u][v] D[v][j]
    s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += sC[g][i * 8 + k] * masks[b * 64 + i * 8 + k] * sD[k * 8 + j];
    __syncthreads();
    sT[g][e] = s;
    __syncthreads();
    s = 0.f;                                                // band[i][j] = sum_u D[u][i] U[u][j]
#pragma unroll
    for (int k = 0; k < 8; ++k) s += sD[k * 8 + i] * sT[g][k * 8 + j];
    if (live && yy < H && xx < W) out[((long long)yy * W + xx) * ldo + ch_off + 3 * b + c] = s * band_scale[b];
  }
}

extern "C" int ff_dct8_bands(const float* x, int C, int H, int W, const float* dct_mat, const float* masks3,
                             const float* band_scale, float* out, int ldo, int ch_off, void* stream) {
  FF_CHECK_ARG(x && dct_mat && masks3 && band_scale && out, "ff_dct8_bands: null pointer");
  FF_CHECK_ARG(C == 3 && H >= 8 && W >= 8, "ff_dct8_bands: expects a 3-channel image of at least 8x8");
  const long long nblk = (long long)C * ((H + 7) / 8) * ((W + 7) / 8);
  hipLaunchKernelGGL(dct8_bands_kernel, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, C, H, W,
                     dct_mat, masks3, band_scale, out, ldo, ch_off);
  FF_LAUNCH_CHECK("ff_dct8_bands");
  return FF_OK;
}

// ------------------------------------------------------------------------------------------ DWT (db4)
// One separable pass: 8-tap lo/hi filters, stride 2, reflect pad 7, along x (axis=1) or y (axis=0).
__global__ __launch_bounds__(256) void dwt_pass_kernel(const float* __restrict__ in, int C, int H, int W, int axis,
                                                       const float* __restrict__ lo, const float* __restrict__ hi,
                                                       float* __restrict__ olo, float* __restrict__ ohi) {
  const int Ho = axis == 0 ? (H + 6) / 2 + 1 : H, Wo = axis == 1 ? (W + 6) / 2 + 1 : W;
  const long long total = (long long)C * Ho * Wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % Wo); long long t = i / Wo;
    const int y = (int)(t % Ho); const int c = (int)(t / Ho);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v;
      if (axis == 1) v = in[((long long)c * H + y) * W + reflect_i(2 * x + k - 7, W)];
      else v = in[((long long)c * H + reflect_i(2 * y + k - 7, H)) * W + x];
      a += lo[k] * v;
      b += hi[k] * v;
    }
    olo[i] = a;
    ohi[i] = b;
  }
}

extern "C" int ff_dwt_pass(const float* in, int C, int H, int W, int axis, const float* lo8, const float* hi8,
                           float* out_lo, float* out_hi, void* stream) {
  FF_CHECK_ARG(in && lo8 && hi8 && out_lo && out_hi && C > 0 && H > 7 && W > 7 && (axis == 0 || axis == 1), "ff_dwt_pass: bad args");
  const int Ho = axis == 0 ? (H + 6) / 2 + 1 : H, Wo = axis == 1 ? (W + 6) / 2 + 1 : W;
  long long nb = ((long long)C * Ho * Wo + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(dwt_pass_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, C, H, W, axis, lo8, hi8,
                     out_lo, out_hi);
  FF_LAUNCH_CHECK("ff_dwt_pass");
  return FF_OK;
}

// ------------------------------------------------------------------------------------------ FFT bands
// rfft2/irfft2 ('ortho') as direct DFTs with host-built twiddle tables tw[n] = (cos, sin)(2 pi n / N):
// 256x256x3 needs ~0.6 GFLOP -- cheaper than any launch-heavy radix pipeline at this size.
// Round 2: the twiddle tables live in LDS (the global-memory version spent its time in 256 dependent L1 round trips per output:
// 505 us for 3 x 256 x 256), the inner loops run four independent partial sums, and the column passes are split over key ranges
// so that a 256 x 129 spectrum gives 200+ workgroups instead of 27.
__global__ __launch_bounds__(256) void rdft_rows_kernel(const float* __restrict__ x, int R, int W, const float* __restrict__ tc,
                                                        const float* __restrict__ ts, float* __restrict__ Y) {
  extern __shared__ float sm[];                            // row[W] | cos[W] | sin[W]
  float* row = sm; float* cs = sm + W; float* sn = sm + 2 * W;
  const int r = blockIdx.x, Wf = W / 2 + 1;
  for (int i = threadIdx.x; i < W; i += 256) { row[i] = x[(long long)r * W + i]; cs[i] = tc[i]; sn[i] = ts[i]; }
  __syncthreads();
  const float nrm = 1.0f / sqrtf((float)W);
  for (int k = threadIdx.x; k < Wf; k += 256) {
    float re[4] = {0.f, 0.f, 0.f, 0.f}, im[4] = {0.f, 0.f, 0.f, 0.f};
    int idx = 0, n = 0;
    for (; n + 3 < W; n += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        re[u] += row[n + u] * cs[idx];
        im[u] -= row[n + u] * sn[idx];
        idx += k;
        if (idx >= W) idx -= W;
      }
    }
    for (; n < W; ++n) {
      re[0] += row[n] * cs[idx]; im[0] -= row[n] * sn[idx];
      idx += k;
      if (idx >= W) idx -= W;
    }
    Y[((long long)r * Wf + k) * 2] = ((re[0] + re[1]) + (re[2] + re[3])) * nrm;
    Y[((long long)r * Wf + k) * 2 + 1] = ((im[0] + im[1]) + (im[2] + im[3])) * nrm;
  }
}

// column DFT over H for CB columns per workgroup and the output rows [blockIdx.z * KH, +KH); sign = -1 forward, +1 inverse.
// With mask_logits the output is multiplied by sigmoid(bilinear(logits)[kh][kw] * temp) (multi_domain_frequency.py:366-374).
__global__ __launch_bounds__(256) void cdft_cols_kernel(const float* __restrict__ Yin, int H, int Wf, int CB, int KH, float sign,
                                                        const float* __restrict__ tc, const float* __restrict__ ts,
                                                        const float* __restrict__ mask_logits, int msz, float temp,
                                                        float* __restrict__ Z) {
  extern __shared__ float sm[];                           // tile [H][CB][2] | cos[H] | sin[H]
  float* tile = sm; float* cs = sm + 2 * H * CB; float* sn = cs + H;
  const int c = blockIdx.y, col0 = blockIdx.x * CB;
  for (int i = threadIdx.x; i < H * CB; i += 256) {
    const int h = i / CB, cc = i % CB;
    const int col = col0 + cc;
    float re = 0.f, im = 0.f;
    if (col < Wf) {
      const long long o = (((long long)c * H + h) * Wf + col) * 2;
      re = Yin[o]; im = Yin[o + 1];
    }
    tile[2 * i] = re; tile[2 * i + 1] = im;
  }
  for (int i = threadIdx.x; i < H; i += 256) { cs[i] = tc[i]; sn[i] = sign * ts[i]; }      // e^{sign * i theta}
  __syncthreads();
  const float nrm = 1.0f / sqrtf((float)H);
  const int cc = threadIdx.x % CB, col = col0 + cc;
  const int kh_end = min(H, (int)(blockIdx.z + 1) * KH);
  for (int kh = blockIdx.z * KH + threadIdx.x / CB; kh < kh_end; kh += 256 / CB) {
    float re[2] = {0.f, 0.f}, im[2] = {0.f, 0.f};
    int idx = 0, h = 0;
    for (; h + 1 < H; h += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float a = tile[2 * ((h + u) * CB + cc)], b = tile[2 * ((h + u) * CB + cc) + 1];
        const float cw = cs[idx], sw = sn[idx];
        re[u] += a * cw - b * sw;
        im[u] += a * sw + b * cw;
        idx += kh;
        if (idx >= H) idx -= H;
      }
    }
    for (; h < H; ++h) {
      const float a = tile[2 * (h * CB + cc)], b = tile[2 * (h * CB + cc) + 1];
      re[0] += a * cs[idx] - b * sn[idx];
      im[0] += a * sn[idx] + b * cs[idx];
      idx += kh;
      if (idx >= H) idx -= H;
    }
    float rr = (re[0] + re[1]) * nrm, ii = (im[0] + im[1]) * nrm;
    if (col < Wf) {
      if (mask_logits) {
        const float sh = (float)msz / (float)H, sw = (float)msz / (float)Wf;
        float fy = fmaxf(sh * ((float)kh + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)col + 0.5f) - 0.5f, 0.f);
        int y0 = min((int)floorf(fy), msz - 1), x0 = min((int)floorf(fx), msz - 1);
        int y1 = min(y0 + 1, msz - 1), x1 = min(x0 + 1, msz - 1);
        float ly = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), lx = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
        if (H == msz) { y0 = y1 = kh; ly = 0.f; }
        if (Wf == msz) { x0 = x1 = col; lx = 0.f; }
        const float lg = (1.f - ly) * ((1.f - lx) * mask_logits[y0 * msz + x0] + lx * mask_logits[y0 * msz + x1]) +
                         ly * ((1.f - lx) * mask_logits[y1 * msz + x0] + lx * mask_logits[y1 * msz + x1]);
        const float m = 1.0f / (1.0f + expf(-lg * temp));
        rr *= m; ii *= m;
      }
      const long long o = (((long long)c * H + kh) * Wf + col) * 2;
      Z[o] = rr; Z[o + 1] = ii;
    }
  }
}

// c2r along W + band write: low = irfft * s_lo -> out[.., ch_lo + c]; high = (x - irfft) * s_hi -> out[.., ch_hi + c]
__global__ __launch_bounds__(256) void irdft_rows_bands_kernel(const float* __restrict__ U, const float* __restrict__ x, int C,
                                                               int H, int W, const float* __restrict__ tc,
                                                               const float* __restrict__ ts, const float* __restrict__ bscale,
                                                               float* __restrict__ out, int ldo, int ch_lo, int ch_hi) {
  extern __shared__ float sm[];                           // urow[Wf][2] | cos[W] | sin[W]
  const int r = blockIdx.x, Wf = W / 2 + 1, c = r / H, y = r % H;
  float* urow = sm; float* cs = sm + 2 * Wf; float* sn = cs + W;
  for (int i = threadIdx.x; i < 2 * Wf; i += 256) urow[i] = U[(long long)r * Wf * 2 + i];
  for (int i = threadIdx.x; i < W; i += 256) { cs[i] = tc[i]; sn[i] = ts[i]; }
  __syncthreads();
  const float nrm = 1.0f / sqrtf((float)W);
  const int kmax = (W - 1) / 2;                           // bins with a conjugate partner
  for (int w = threadIdx.x; w < W; w += 256) {
    float s0 = urow[0], s1 = 0.f;
    if ((W & 1) == 0) s0 += ((w & 1) ? -1.f : 1.f) * urow[2 * (W / 2)];
    int idx = 0, k = 1;
    for (; k + 1 <= kmax; k += 2) {
      idx += w;
      if (idx >= W) idx -= W;
      s0 += 2.f * (urow[2 * k] * cs[idx] - urow[2 * k + 1] * sn[idx]);
      idx += w;
      if (idx >= W) idx -= W;
      s1 += 2.f * (urow[2 * k + 2] * cs[idx] - urow[2 * k + 3] * sn[idx]);
    }
    for (; k <= kmax; ++k) {
      idx += w;
      if (idx >= W) idx -= W;
      s0 += 2.f * (urow[2 * k] * cs[idx] - urow[2 * k + 1] * sn[idx]);
    }
    const float s = (s0 + s1) * nrm;
    const float xv = x[(long long)r * W + w];
    float* o = out + ((long long)y * W + w) * ldo;
    o[ch_lo + c] = s * bscale[0];
    o[ch_hi + c] = (xv - s) * bscale[1];
  }
}

extern "C" int ff_fft_bands(const float* x, int C, int H, int W, const float* twW_cos, const float* twW_sin,
                            const float* twH_cos, const float* twH_sin, const float* mask_logits, int msz, float temp,
                            const float* band_scale2, float* work, long long work_floats, float* out, int ldo, int ch_lo,
                            int ch_hi, void* stream) {
  FF_CHECK_ARG(x && twW_cos && twW_sin && twH_cos && twH_sin && mask_logits && band_scale2 && work && out, "ff_fft_bands: null pointer");
  FF_CHECK_ARG(C > 0 && H > 1 && W > 1 && W <= 8192 && H <= 8192, "ff_fft_bands: bad dims");
  const int Wf = W / 2 + 1;
  const long long spec = (long long)C * H * Wf * 2;
  FF_CHECK_ARG(work_floats >= 2 * spec, "ff_fft_bands: workspace too small (need %lld floats)", 2 * spec);
  float* Y = work;
  float* Z = work + spec;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rdft_rows_kernel, dim3(C * H), dim3(256), (size_t)W * 12, st, x, C * H, W, twW_cos, twW_sin, Y);
  const int CB = H <= 512 ? 16 : 4;
  const size_t clds = (size_t)H * CB * 8 + (size_t)H * 8;
  FF_CHECK_ARG(clds <= 64 * 1024, "ff_fft_bands: H too large for the column tile");
  // split the output rows of a column block over workgroups until the grid has a few per CU
  const int ncb = (Wf + CB - 1) / CB;
  int ksplit = 1;
  while (ncb * C * ksplit < 512 && (H + ksplit - 1) / ksplit > 256 / CB) ksplit *= 2;
  const int KH = (H + ksplit - 1) / ksplit;
  dim3 gc(ncb, C, (H + KH - 1) / KH);
  hipLaunchKernelGGL(cdft_cols_kernel, gc, dim3(256), clds, st, Y, H, Wf, CB, KH, -1.f, twH_cos, twH_sin, mask_logits,
                     msz, temp, Z);
  hipLaunchKernelGGL(cdft_cols_kernel, gc, dim3(256), clds, st, Z, H, Wf, CB, KH, 1.f, twH_cos, twH_sin,
                     (const float*)nullptr, 0, 0.f, Y);
  hipLaunchKernelGGL(irdft_rows_bands_kernel, dim3(C * H), dim3(256), (size_t)Wf * 8 + (size_t)W * 8, st, Y, x, C, H, W, twW_cos, twW_sin,
                     band_scale2, out, ldo, ch_lo, ch_hi);
  FF_LAUNCH_CHECK("ff_fft_bands");
  return FF_OK;
}

// ------------------------------------------------------------------------------------------ plain rfft2 / irfft2 ('ortho')
// The two halves of ff_fft_bands as separate entry points, for the training path (csrc/train_ops.hip, isr2_amd/autograd.py): the
// learnable mask is applied between them by differentiable host-sequenced kernels, and the backward of irfft2 is an rfft2.
//   ff_rfft2 : planes [C][H][W] -> spec [C][H][Wf][2];  ff_irfft2 : spec -> planes.  work: C*H*Wf*2 floats each.
__global__ __launch_bounds__(256) void irdft_rows_kernel(const float* __restrict__ U, int H, int W, const float* __restrict__ tc,
                                                         const float* __restrict__ ts, float* __restrict__ out) {
  extern __shared__ float sm[];                           // urow[Wf][2] | cos[W] | sin[W]
  const int r = blockIdx.x, Wf = W / 2 + 1;
  float* urow = sm; float* cs = sm + 2 * Wf; float* sn = cs + W;
  for (int i = threadIdx.x; i < 2 * Wf; i += 256) urow[i] = U[(long long)r * Wf * 2 + i];
  for (int i = threadIdx.x; i < W; i += 256) { cs[i] = tc[i]; sn[i] = ts[i]; }
  __syncthreads();
  const float nrm = 1.0f / sqrtf((float)W);
  const int kmax = (W - 1) / 2;
  for (int w = threadIdx.x; w < W; w += 256) {
    float s0 = urow[0], s1 = 0.f;
    if ((W & 1) == 0) s0 += ((w & 1) ? -1.f : 1.f) * urow[2 * (W / 2)];
    int idx = 0, k = 1;
    for (; k + 1 <= kmax; k += 2) {
      idx += w;
      if (idx >= W) idx -= W;
      s0 += 2.f * (urow[2 * k] * cs[idx] - urow[2 * k + 1] * sn[idx]);
      idx += w;
      if (idx >= W) idx -= W;
      s1 += 2.f * (urow[2 * k + 2] * cs[idx] - urow[2 * k + 3] * sn[idx]);
    }
    for (; k <= kmax; ++k) {
      idx += w;
      if (idx >= W) idx -= W;
      s0 += 2.f * (urow[2 * k] * cs[idx] - urow[2 * k + 1] * sn[idx]);
    }
    out[(long long)r * W + w] = (s0 + s1) * nrm;
  }
}

static int fft2_cols(const float* in, float* out, int C, int H, int Wf, float sign, const float* twH_cos, const float* twH_sin, hipStream_t st) {
  const int CB = H <= 512 ? 16 : 4;
  const size_t clds = (size_t)H * CB * 8 + (size_t)H * 8;
  if (clds > 64 * 1024) { ff_set_error("fft2: H too large for the column tile"); return FF_ERR_ARG; }
  const int ncb = (Wf + CB - 1) / CB;
  int ksplit = 1;
  while (ncb * C * ksplit < 512 && (H + ksplit - 1) / ksplit > 256 / CB) ksplit *= 2;
  const int KH = (H + ksplit - 1) / ksplit;
  dim3 gc(ncb, C, (H + KH - 1) / KH);
  hipLaunchKernelGGL(cdft_cols_kernel, gc, dim3(256), clds, st, in, H, Wf, CB, KH, sign, twH_cos, twH_sin, (const float*)nullptr, 0, 0.f, out);
  return FF_OK;
}

extern "C" int ff_rfft2(const float* x, int C, int H, int W, const float* twW_cos, const float* twW_sin, const float* twH_cos,
                        const float* twH_sin, float* work, long long work_floats, float* spec, void* stream) {
  FF_CHECK_ARG(x && twW_cos && twW_sin && twH_cos && twH_sin && work && spec, "ff_rfft2: null pointer");
  FF_CHECK_ARG(C > 0 && C <= 65535 && H > 1 && W > 1 && W <= 8192 && H <= 8192, "ff_rfft2: bad dims");
  const int Wf = W / 2 + 1;
  FF_CHECK_ARG(work_floats >= (long long)C * H * Wf * 2, "ff_rfft2: workspace too small (need %lld floats)", (long long)C * H * Wf * 2);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rdft_rows_kernel, dim3(C * H), dim3(256), (size_t)W * 12, st, x, C * H, W, twW_cos, twW_sin, work);
  if (int rc = fft2_cols(work, spec, C, H, Wf, -1.f, twH_cos, twH_sin, st)) return rc;
  FF_LAUNCH_CHECK("ff_rfft2");
  return FF_OK;
}

extern "C" int ff_irfft2(const float* spec, int C, int H, int W, const float* twW_cos, const float* twW_sin, const float* twH_cos,
                         const float* twH_sin, float* work, long long work_floats, float* out, void* stream) {
  FF_CHECK_ARG(spec && twW_cos && twW_sin && twH_cos && twH_sin && work && out, "ff_irfft2: null pointer");
  FF_CHECK_ARG(C > 0 && C <= 65535 && H > 1 && W > 1 && W <= 8192 && H <= 8192, "ff_irfft2: bad dims");
  const int Wf = W / 2 + 1;
  FF_CHECK_ARG(work_floats >= (long long)C * H * Wf * 2, "ff_irfft2: workspace too small (need %lld floats)", (long long)C * H * Wf * 2);
  hipStream_t st = (hipStream_t)stream;
  if (int rc = fft2_cols(spec, work, C, H, Wf, 1.f, twH_cos, twH_sin, st)) return rc;
  hipLaunchKernelGGL(irdft_rows_kernel, dim3(C * H), dim3(256), (size_t)Wf * 8 + (size_t)W * 8, st, work, H, W, twW_cos, twW_sin, out);
  FF_LAUNCH_CHECK("ff_irfft2");
  return FF_OK;
}

// Device PSNR / SSIM evaluator: the reference's src/utils/metrics.py on HIP, so parity and quality numbers
// never leave the device.
//   rgb_to_y              metrics.py:30-52   y = (65.481 r + 128.553 g + 24.966 b + 16) / 255  (ITU-R BT.601)
//   calculate_psnr        metrics.py:76-126  clamp to [0,1], crop the border, optional Y, mean squared error
//   calculate_ssim_torch  metrics.py:129-190 11x11 Gaussian (sigma 1.5) means / variances with ZERO padding, SSIM map, mean
// Both kernels produce per-workgroup partial sums in double precision that ff_metric_finish adds in a fixed order
// (deterministic; a float32 mean over 10^6-10^7 pixels would lose digits the oracle comparison is sensitive to).
// Images are planar NCHW fp32 [C][H][W] (C = 3 or 1), the layout the plugin's outputs have.
#include "ff_common.h"

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// channel value at (c, y, x) of the evaluated image: clamped, optionally converted to luma (then c == 0)
__device__ __forceinline__ float metric_px(const float* __restrict__ img, int C, int H, int W, int c, int y, int x, int use_y) {
  const long long P = (long long)H * W, o = (long long)y * W + x;
  if (use_y && C == 3) {
    const float r = clamp01(img[o]), g = clamp01(img[P + o]), b = clamp01(img[2 * P + o]);
    return (65.481f * r + 128.553f * g + 24.966f * b + 16.0f) / 255.0f;
  }
  return clamp01(img[(long long)c * P + o]);
}

__device__ __forceinline__ double block_sum_256(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void psnr_sse_kernel(const float* __restrict__ a, const float* __restrict__ b, int C, int H, int W,
                                                       int crop, int use_y, double* __restrict__ partial) {
  __shared__ double sh[4];
  const int Hc = H - 2 * crop, Wc = W - 2 * crop, Ce = (use_y && C == 3) ? 1 : C;
  const long long total = (long long)Ce * Hc * Wc;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % Wc) + crop, y = (int)((i / Wc) % Hc) + crop, c = (int)(i / ((long long)Wc * Hc));
    const float d = metric_px(a, C, H, W, c, y, x, use_y) - metric_px(b, C, H, W, c, y, x, use_y);
    acc += (double)(d * d);
  }
  const double s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// One workgroup = a 16x16 tile of SSIM-map pixels of one channel: the 26x26 input patches of both images go to LDS (zero outside
// the CROPPED image, as F.conv2d's zero padding does), then a separable 11-tap Gaussian of x, y, x^2, y^2, xy.
#define SS_T 16
#define SS_R 5
#define SS_P (SS_T + 2 * SS_R)
__global__ __launch_bounds__(256) void ssim_map_kernel(const float* __restrict__ a, const float* __restrict__ b, int C, int H, int W,
                                                       int crop, int use_y, const float* __restrict__ gw, double* __restrict__ partial) {
  __shared__ float pa[SS_P][SS_P + 1], pb[SS_P][SS_P + 1];
  __shared__ float hq[5][SS_P][SS_T + 1];
  __shared__ double sh[4];
  __shared__ float g[11];
  const int Hc = H - 2 * crop, Wc = W - 2 * crop;
  const int tx = blockIdx.x, ty = blockIdx.y, c = blockIdx.z;
  const int tid = threadIdx.x;
  if (tid < 11) g[tid] = gw[tid];
  for (int i = tid; i < SS_P * SS_P; i += 256) {
    const int ly = i / SS_P, lx = i % SS_P;
    const int y = ty * SS_T + ly - SS_R, x = tx * SS_T + lx - SS_R;
    float va = 0.f, vb = 0.f;
    if (y >= 0 && y < Hc && x >= 0 && x < Wc) {
      va = metric_px(a, C, H, W, c, y + crop, x + crop, use_y);
      vb = metric_px(b, C, H, W, c, y + crop, x + crop, use_y);
    }
    pa[ly][lx] = va;
    pb[ly][lx] = vb;
  }
  __syncthreads();
  for (int i = tid; i < SS_P * SS_T; i += 256) {
    const int ly = i / SS_T, lx = i % SS_T;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float va = pa[ly][lx + k], vb = pb[ly][lx + k], wk = g[k];
      s0 += wk * va; s1 += wk * vb; s2 += wk * va * va; s3 += wk * vb * vb; s4 += wk * va * vb;
    }
    hq[0][ly][lx] = s0; hq[1][ly][lx] = s1; hq[2][ly][lx] = s2; hq[3][ly][lx] = s3; hq[4][ly][lx] = s4;
  }
  __syncthreads();
  const int ly = tid / SS_T, lx = tid % SS_T;
  const int oy = ty * SS_T + ly, ox = tx * SS_T + lx;
  double v = 0.0;
  if (oy < Hc && ox < Wc) {
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float wk = g[k];
      m1 += wk * hq[0][ly + k][lx]; m2 += wk * hq[1][ly + k][lx];
      e11 += wk * hq[2][ly + k][lx]; e22 += wk * hq[3][ly + k][lx]; e12 += wk * hq[4][ly + k][lx];
    }
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
    const float s11 = e11 - m11, s22 = e22 - m22, s12 = e12 - m12;
    v = (double)(((2.f * m12 + C1) * (2.f * s12 + C2)) / ((m11 + m22 + C1) * (s11 + s22 + C2)));
  }
  const double s = block_sum_256(v, sh);
  if (tid == 0) partial[((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void metric_finish_kernel(const double* __restrict__ partial, int n, double scale, double* __restrict__ out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];      // fixed assignment of partials to threads: deterministic
  const double s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) out[0] = s * scale;
}

static int metric_dims_ok(int C, int H, int W, int crop) { return (C == 1 || C == 3) && H > 2 * crop && W > 2 * crop && crop >= 0; }

extern "C" int ff_metric_workspace(int C, int H, int W, int crop) {
  if (!metric_dims_ok(C, H, W, crop)) return -1;
  const int Hc = H - 2 * crop, Wc = W - 2 * crop;
  const long long ss = (long long)ff_cdiv(Wc, SS_T) * ff_cdiv(Hc, SS_T) * C;
  return (int)(ss > 1024 ? ss : 1024);                               // doubles
}

// mean squared error of (clamped, cropped, optionally luma) a vs b -> out_mse[0] (device double)
extern "C" int ff_psnr_mse(const float* a, const float* b, int C, int H, int W, int crop, int use_y, double* work, int nwork,
                           double* out_mse, void* stream) {
  FF_CHECK_ARG(a && b && work && out_mse && metric_dims_ok(C, H, W, crop), "ff_psnr_mse: bad args");
  FF_CHECK_ARG(nwork >= 1024, "ff_psnr_mse: workspace too small");
  const int Ce = (use_y && C == 3) ? 1 : C;
  const long long total = (long long)Ce * (H - 2 * crop) * (W - 2 * crop);
  int nb = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(psnr_sse_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, C, H, W, crop, use_y, work);
  hipLaunchKernelGGL(metric_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, work, nb, 1.0 / (double)total, out_mse);
  FF_LAUNCH_CHECK("ff_psnr_mse");
  return FF_OK;
}

// mean of the SSIM map (11x11 Gaussian window given as its 11 separable weights gauss11) -> out_ssim[0] (device double)
extern "C" int ff_ssim_mean(const float* a, const float* b, int C, int H, int W, int crop, int use_y, const float* gauss11,
                            double* work, int nwork, double* out_ssim, void* stream) {
  FF_CHECK_ARG(a && b && gauss11 && work && out_ssim && metric_dims_ok(C, H, W, crop), "ff_ssim_mean: bad args");
  const int Ce = (use_y && C == 3) ? 1 : C, Hc = H - 2 * crop, Wc = W - 2 * crop;
  const dim3 grid(ff_cdiv(Wc, SS_T), ff_cdiv(Hc, SS_T), Ce);
  const long long nb = (long long)grid.x * grid.y * grid.z;
  FF_CHECK_ARG(nb <= nwork && grid.y < 65536, "ff_ssim_mean: workspace too small / image too large");
  hipLaunchKernelGGL(ssim_map_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, b, C, H, W, crop, use_y, gauss11, work);
  hipLaunchKernelGGL(metric_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, work, (int)nb, 1.0 / ((double)Ce * Hc * Wc), out_ssim);
  FF_LAUNCH_CHECK("ff_ssim_mean");
  return FF_OK;
}

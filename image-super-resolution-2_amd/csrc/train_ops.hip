// Training-side primitives of the fusion-only training step (SURVEY 8f rank 1 / BASELINE config 5; reference
// train.py:308-356 `train_epoch_cached`): the backward halves of the fusion stack's operators plus the small differentiable
// tensor vocabulary the host (isr2_amd/autograd.py) composes the pointwise formulas from.  Everything is fp32 and
// DETERMINISTIC: every reduction is two-stage with a fixed order (no atomics), so a training step is bit-reproducible.
//
//   ff_ew_fma / ff_ew_unary          broadcast multiply-add and activations (+ their derivative forms)
//   ff_reduce_cols / ff_reduce_rows  sums over pixels (per image group) / over channels, optionally of a product x*y
//   ff_conv2d_wgrad                  dW of nn.Conv2d / nn.Linear on the fp32 matrix cores (v_mfma_f32_32x32x2_f32)
//   ff_conv_weight_flipT             W [Co][taps][Ci] -> [Ci][flipped taps][Co]: the data gradient is a forward conv with it
//   ff_dwconv2d_wgrad, ff_taps_reverse   depth-wise weight gradient; reversed taps for the depth-wise data gradient
//   ff_resize_bilinear_adj, ff_avgpool2_adj   adjoints of F.interpolate(bilinear, align_corners=False) / avg_pool2d(2)
//   ff_layernorm_bwd                 nn.LayerNorm backward (dx + dgamma / dbeta)
//   ff_bn_train_*                    nn.BatchNorm2d in training mode: batch statistics per group of images, running-stat
//                                    update (momentum 0.1, unbiased variance), backward
//   ff_band_mha_train / _bwd         the 9-token (bands) / 3-token (experts) per-pixel attention core with dropout on the
//                                    attention weights (nn.MultiheadAttention, large_kernel_attention.py:196,296) and its backward
//   ff_dynamic_gates_bwd             fusion_network.py:226-234 backward
//   ff_permute_rows                  [A][B][C] -> [B][A][C]
//   ff_spec_mask_mul / _grad         X * m on a half spectrum and d m = sum Re(gY conj X) (multi_domain_frequency.py:366-385)
//   ff_l1_loss_grad, ff_grad_sqnorm, ff_adamw_ema_step   L1 (perceptual_loss.py:86-105), clip_grad_norm_ + AdamW + EMA
//                                    (train.py:338-351, checkpoint_manager.py:400-407) over one flat parameter buffer
#include "ff_common.h"

__device__ __forceinline__ float tw_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------------------------------------------------------
// Broadcast operand: kind 0 full [rows][ld], 1 per row (element r*ld), 2 per (group, column) [G][C] with group = r / rpg,
// 3 one device scalar, 4 absent.
struct EwOperand { const float* p; int ld; int kind; };
__device__ __forceinline__ float ew_get(const EwOperand& o, long long r, int c, int C, long long rpg, float absent) {
  switch (o.kind) {
    case 0: return o.p[r * o.ld + c];
    case 1: return o.p[r * o.ld];
    case 2: return o.p[(r / rpg) * C + c];
    case 3: return o.p[0];
    default: return absent;
  }
}

// out = a * b + c_scale * c      (a NULL -> 1, b absent -> 1, c absent -> 0; clamp01 optional)
__global__ __launch_bounds__(256) void ew_fma_kernel(float* __restrict__ out, int ldo, const float* __restrict__ a, int lda,
                                                     EwOperand b, EwOperand c, float c_scale, long long rows, int C,
                                                     long long rpg, int clamp01) {
  const long long total = rows * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C;
    const int col = (int)(i - r * C);
    float v = (a ? a[r * lda + col] : 1.f) * ew_get(b, r, col, C, rpg, 1.f);
    if (c.kind != 4) v += c_scale * ew_get(c, r, col, C, rpg, 0.f);
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    out[r * ldo + col] = v;
  }
}

extern "C" int ff_ew_fma(float* out, int ldo, const float* a, int lda, const float* b, int ldb, int b_kind, const float* c,
                         int ldc, int c_kind, float c_scale, long long rows, int C, long long rows_per_group, int clamp01,
                         void* stream) {
  FF_CHECK_ARG(out && rows > 0 && C > 0 && ldo >= C && (!a || lda >= C), "ff_ew_fma: bad args");
  FF_CHECK_ARG(b_kind >= 0 && b_kind <= 4 && c_kind >= 0 && c_kind <= 4, "ff_ew_fma: operand kind must be 0..4");
  FF_CHECK_ARG((b_kind == 4 || b) && (c_kind == 4 || c), "ff_ew_fma: operand pointer missing");
  FF_CHECK_ARG((b_kind != 0 || ldb >= C) && (c_kind != 0 || ldc >= C), "ff_ew_fma: full operand with ld < C");
  FF_CHECK_ARG((b_kind != 2 && c_kind != 2) || (rows_per_group > 0 && rows % rows_per_group == 0), "ff_ew_fma: rows_per_group must divide rows");
  long long nb = (rows * C + 255) / 256;
  if (nb > 16384) nb = 16384;
  EwOperand ob{b, ldb, b_kind}, oc{c, ldc, c_kind};
  hipLaunchKernelGGL(ew_fma_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, out, ldo, a, lda, ob, oc, c_scale,
                     rows, C, rows_per_group > 0 ? rows_per_group : rows, clamp01);
  FF_LAUNCH_CHECK("ff_ew_fma");
  return FF_OK;
}

// unary ops and their derivative forms.  op < 16: out = f(x) ; op >= 16: out = g * f'(x) (or from the output y where noted)
enum { U_GELU = 0, U_RELU = 1, U_SIGMOID = 2, U_SOFTPLUS = 3, U_ABS = 4, U_CLAMP01 = 5, U_SCALE = 6, U_DIV_EPS = 7, U_CLAMP_MIN = 8, U_EXP = 9,
       U_GELU_BWD = 16, U_RELU_BWD = 17, U_SIGMOID_BWD_Y = 18, U_SOFTPLUS_BWD = 19, U_ABS_BWD = 20, U_CLAMP01_BWD = 21,
       U_RECIP_BWD = 22, U_CLAMP_MIN_BWD = 23, U_EXP_BWD_Y = 24 };

__device__ __forceinline__ float ew_unary_apply(int op, float x, float g, float p0) {
  switch (op) {
    case U_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    case U_RELU: return x > 0.f ? x : 0.f;
    case U_SIGMOID: return 1.0f / (1.0f + expf(-x));
    case U_SOFTPLUS: return x > 20.f ? x : log1pf(expf(x));                      // F.softplus(beta 1, threshold 20)
    case U_ABS: return fabsf(x);
    case U_CLAMP01: return fminf(fmaxf(x, 0.f), 1.f);
    case U_SCALE: return x * p0;
    case U_DIV_EPS: return 1.0f / (x + p0);                                      // 1 / (x + eps)
    case U_CLAMP_MIN: return fmaxf(x, p0);
    case U_EXP: return expf(x);
    case U_GELU_BWD: {                                                            // d/dx [x Phi(x)] = Phi(x) + x phi(x)
      const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
      return g * (cdf + x * pdf);
    }
    case U_RELU_BWD: return x > 0.f ? g : 0.f;
    case U_SIGMOID_BWD_Y: return g * x * (1.0f - x);                              // x = sigmoid output
    case U_SOFTPLUS_BWD: return x > 20.f ? g : g / (1.0f + expf(-x));
    case U_ABS_BWD: return x > 0.f ? g : (x < 0.f ? -g : 0.f);                    // torch: sgn(0) = 0
    case U_CLAMP01_BWD: return (x >= 0.f && x <= 1.f) ? g : 0.f;                  // torch.clamp: bounds inclusive
    case U_RECIP_BWD: return -g * x * x;                                          // x = the reciprocal y = 1/(u+eps): dy/du = -y^2
    case U_CLAMP_MIN_BWD: return x >= p0 ? g : 0.f;
    case U_EXP_BWD_Y: return g * x;                                               // x = exp output
    default: return x;
  }
}

__global__ __launch_bounds__(256) void ew_unary_kernel(int op, const float* __restrict__ x, int ldx, const float* __restrict__ g,
                                                       int ldg, float* __restrict__ out, int ldo, long long rows, int C, float p0) {
  const long long total = rows * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C;
    const int c = (int)(i - r * C);
    out[r * ldo + c] = ew_unary_apply(op, x[r * ldx + c], g ? g[r * ldg + c] : 0.f, p0);
  }
}

extern "C" int ff_ew_unary(int op, const float* x, int ldx, const float* g, int ldg, float* out, int ldo, long long rows, int C,
                           float p0, void* stream) {
  FF_CHECK_ARG(x && out && rows > 0 && C > 0 && ldx >= C && ldo >= C, "ff_ew_unary: bad args");
  FF_CHECK_ARG((op >= 0 && op <= 9) || (op >= 16 && op <= 24), "ff_ew_unary: unknown op %d", op);
  FF_CHECK_ARG(op < 16 || (g && ldg >= C), "ff_ew_unary: derivative forms need the incoming gradient");
  long long nb = (rows * C + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(ew_unary_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, op, x, ldx, g, ldg, out, ldo, rows, C, p0);
  FF_LAUNCH_CHECK("ff_ew_unary");
  return FF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Column sums per group of rows:  out[g][c] = scale * sum_{r in group g} x[r][c] (* y[r][c]) (* yrow[r])
// stage 1: grid (chunks, G); stage 2: fixed-order sum over the chunks.
__global__ __launch_bounds__(256) void reduce_cols_stage1(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                                          int y_kind, long long rpg, int C, int rows_per_chunk,
                                                          float* __restrict__ part) {
  __shared__ float red[256];
  const int chunk = blockIdx.x, g = blockIdx.y, nch = gridDim.x;
  const long long r0 = (long long)g * rpg + (long long)chunk * rows_per_chunk;
  long long r1 = r0 + rows_per_chunk;
  const long long rend = (long long)(g + 1) * rpg;
  if (r1 > rend) r1 = rend;
  float* o = part + ((long long)g * nch + chunk) * C;
  if (C <= 64) {
    int Cp = 1;
    while (Cp < C) Cp <<= 1;
    const int R = 256 / Cp, col = threadIdx.x % Cp, rl = threadIdx.x / Cp;
    float s0 = 0.f, s1 = 0.f;
    if (col < C) {
      long long r = r0 + rl;
      for (; r + R < r1; r += 2 * R) {
        float v0 = x[r * ldx + col], v1 = x[(r + R) * ldx + col];
        if (y_kind == 0) { v0 *= y[r * ldy + col]; v1 *= y[(r + R) * ldy + col]; }
        else if (y_kind == 1) { v0 *= y[r * ldy]; v1 *= y[(r + R) * ldy]; }
        s0 += v0; s1 += v1;
      }
      if (r < r1) {
        float v0 = x[r * ldx + col];
        if (y_kind == 0) v0 *= y[r * ldy + col];
        else if (y_kind == 1) v0 *= y[r * ldy];
        s0 += v0;
      }
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    for (int h = R / 2; h > 0; h >>= 1) {
      if (rl < h) red[threadIdx.x] += red[threadIdx.x + h * Cp];
      __syncthreads();
    }
    if (rl == 0 && col < C) o[col] = red[col];
  } else {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int col = c0 + lane;
      float s0 = 0.f, s1 = 0.f;
      if (col < C) {
        long long r = r0 + w;
        for (; r + 4 < r1; r += 8) {
          float v0 = x[r * ldx + col], v1 = x[(r + 4) * ldx + col];
          if (y_kind == 0) { v0 *= y[r * ldy + col]; v1 *= y[(r + 4) * ldy + col]; }
          else if (y_kind == 1) { v0 *= y[r * ldy]; v1 *= y[(r + 4) * ldy]; }
          s0 += v0; s1 += v1;
        }
        if (r < r1) {
          float v0 = x[r * ldx + col];
          if (y_kind == 0) v0 *= y[r * ldy + col];
          else if (y_kind == 1) v0 *= y[r * ldy];
          s0 += v0;
        }
      }
      red[threadIdx.x] = s0 + s1;
      __syncthreads();
      if (w == 0 && col < C) o[col] = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
      __syncthreads();
    }
  }
}

// stage 2: out[g][c] = scale * sum_k part[g][k][c].  A workgroup owns 16 columns of one group; its 16 thread rows take the chunks
// k = row, row + 16, ... with four independent partial sums each (a single thread walking thousands of chunks was latency bound:
// 64 us per launch, 12 % of a training step), then a fixed-order LDS tree.
__global__ __launch_bounds__(256) void reduce_cols_stage2(const float* __restrict__ part, int nch, int C, int G, float scale,
                                                          float* __restrict__ out, int accumulate) {
  __shared__ float red[16][17];
  const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + tc, g = blockIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    const float* p = part + (long long)g * nch * C + c;
    int k = tr;
    for (; k + 48 < nch; k += 64) {
      s0 += p[(long long)k * C]; s1 += p[(long long)(k + 16) * C]; s2 += p[(long long)(k + 32) * C]; s3 += p[(long long)(k + 48) * C];
    }
    for (; k < nch; k += 16) s0 += p[(long long)k * C];
  }
  red[tr][tc] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int h = 8; h > 0; h >>= 1) {
    if (tr < h) red[tr][tc] += red[tr + h][tc];
    __syncthreads();
  }
  if (tr == 0 && c < C) {
    const float v = red[0][tc] * scale;
    const long long e = (long long)g * C + c;
    out[e] = accumulate ? out[e] + v : v;
  }
}

static void launch_reduce_stage2(const float* part, int nch, int C, int G, float scale, float* out, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(reduce_cols_stage2, dim3((C + 15) / 16, G), dim3(256), 0, st, part, nch, C, G, scale, out, accumulate);
}

static int reduce_cols_chunks(long long rpg, long long G) {
  long long nch = (rpg + 511) / 512;
  long long cap = 1024 / (G > 0 ? G : 1);
  if (cap < 1) cap = 1;
  if (nch > cap) nch = cap;
  if (nch < 1) nch = 1;
  return (int)nch;
}

extern "C" long long ff_reduce_cols_workspace(long long rows, int C, long long rows_per_group) {
  if (rows <= 0 || C <= 0) return -1;
  const long long rpg = rows_per_group > 0 ? rows_per_group : rows;
  const long long G = rows / rpg;
  return G * reduce_cols_chunks(rpg, G) * (long long)C;
}

extern "C" int ff_reduce_cols(const float* x, int ldx, const float* y, int ldy, int y_kind, long long rows, int C,
                              long long rows_per_group, float scale, float* out, int accumulate, float* work,
                              long long work_floats, void* stream) {
  FF_CHECK_ARG(x && out && work && rows > 0 && C > 0 && ldx >= C, "ff_reduce_cols: bad args");
  FF_CHECK_ARG(y_kind == 4 || (y && (y_kind == 0 || y_kind == 1)), "ff_reduce_cols: y_kind must be 0 (full), 1 (per row) or 4 (absent)");
  FF_CHECK_ARG(y_kind != 0 || ldy >= C, "ff_reduce_cols: y with ld < C");
  const long long rpg = rows_per_group > 0 ? rows_per_group : rows;
  FF_CHECK_ARG(rows % rpg == 0, "ff_reduce_cols: rows_per_group must divide rows");
  const long long G = rows / rpg;
  FF_CHECK_ARG(G <= 65535, "ff_reduce_cols: too many groups");
  const int nch = reduce_cols_chunks(rpg, G);
  FF_CHECK_ARG(work_floats >= G * nch * (long long)C, "ff_reduce_cols: workspace too small (need %lld floats)", G * nch * (long long)C);
  const int rpc = (int)((rpg + nch - 1) / nch);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(reduce_cols_stage1, dim3(nch, (unsigned)G), dim3(256), 0, st, x, ldx, y, ldy, y_kind, rpg, C, rpc, work);
  launch_reduce_stage2(work, nch, C, (int)G, scale, out, accumulate, st);
  FF_LAUNCH_CHECK("ff_reduce_cols");
  return FF_OK;
}

// Row sums: out[r*ldo] = scale * sum_c x[r][c] (* y[r][c] | * ycol[c])
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                                          int y_kind, long long rows, int C, float scale, float* __restrict__ out,
                                                          int ldo, int accumulate) {
  if (C <= 16) {
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
      float s = 0.f;
      for (int c = 0; c < C; ++c) {
        float v = x[r * ldx + c];
        if (y_kind == 0) v *= y[r * ldy + c];
        else if (y_kind == 2) v *= y[c];
        s += v;
      }
      s *= scale;
      out[r * ldo] = accumulate ? out[r * ldo] + s : s;
    }
  } else {
    const int lane = threadIdx.x & 63;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
      float s = 0.f;
      for (int c = lane; c < C; c += 64) {
        float v = x[r * ldx + c];
        if (y_kind == 0) v *= y[r * ldy + c];
        else if (y_kind == 2) v *= y[c];
        s += v;
      }
      s = tw_wave_sum(s) * scale;
      if (lane == 0) out[r * ldo] = accumulate ? out[r * ldo] + s : s;
    }
  }
}

extern "C" int ff_reduce_rows(const float* x, int ldx, const float* y, int ldy, int y_kind, long long rows, int C, float scale,
                              float* out, int ldo, int accumulate, void* stream) {
  FF_CHECK_ARG(x && out && rows > 0 && C > 0 && ldx >= C && ldo >= 1, "ff_reduce_rows: bad args");
  FF_CHECK_ARG(y_kind == 4 || (y && (y_kind == 0 || y_kind == 2)), "ff_reduce_rows: y_kind must be 0 (full), 2 (per column) or 4 (absent)");
  long long nb = C <= 16 ? (rows + 255) / 256 : (rows + 3) / 4;
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, y_kind, rows, C,
                     scale, out, ldo, accumulate);
  FF_LAUNCH_CHECK("ff_reduce_rows");
  return FF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of a stride-1 convolution / linear layer on the fp32 matrix cores:
//   dW[co][tap*Cin + ci] = sum_{b,y,x} dz[b,y,x,co] * in[b, y+ky-py, x+kx-px, ci]           (zero padding)
// A job = (64 output channels, 32 input channels, one tap); a workgroup (4 waves) walks the image rows of its chunk, each wave
// keeping two 32x32 accumulators: D[m = co][n = ci] += A[m][k = pixel] * B[k][n].  Both operands are read straight from the NHWC
// tensors -- for a pixel pair the 64 lanes read two 128-byte channel segments -- so no LDS staging is needed.
// Stage 2 sums the per-chunk partial tiles in a fixed order.
template <int MI>
__global__ __launch_bounds__(256) void conv_wgrad_stage1(const float* __restrict__ in, int ldx, const float* __restrict__ dz, int ldz,
                                                         int B, int H, int W, int Cin, int Cout, int KH, int KW, int py, int px,
                                                         int n_ci_t, int rows_per_chunk, float* __restrict__ part) {
  __shared__ float red[4][MI * 1024];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, hh = lane >> 5;
  const int taps = KH * KW;
  int job = blockIdx.x;
  const int tap = job % taps; job /= taps;
  const int ci_t = job % n_ci_t; const int co_t = job / n_ci_t;
  const int ky = tap / KW, kx = tap - ky * KW;
  const int co0 = co_t * (32 * MI), ci0 = ci_t * 32;
  const int ci = ci0 + l31;
  const bool ci_ok = ci < Cin;
  bool co_ok[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) co_ok[i] = co0 + 32 * i + l31 < Cout;
  f32x16 acc[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const long long nrows = (long long)B * H;
  const long long row0 = (long long)blockIdx.y * rows_per_chunk;
  long long row1 = row0 + rows_per_chunk;
  if (row1 > nrows) row1 = nrows;
  for (long long row = row0 + wid; row < row1; row += 4) {
    const int y = (int)(row % H);
    const int iy = y + ky - py;
    if ((unsigned)iy >= (unsigned)H) continue;                        // the whole row of this tap reads padding
    const long long b = row / H;
    const float* zrow = dz + row * W * (long long)ldz + co0 + l31;
    const float* xrow = in + ((b * H + iy) * (long long)W) * ldx + ci;
    for (int xb = 0; xb < W; xb += 8) {
      float av[MI][4], bv[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int x = xb + 2 * s + hh;
        const int ix = x + kx - px;
        const bool xok = x < W;
#pragma unroll
        for (int i = 0; i < MI; ++i) av[i][s] = (xok && co_ok[i]) ? zrow[(long long)x * ldz + 32 * i] : 0.f;
        bv[s] = (xok && ci_ok && (unsigned)ix < (unsigned)W) ? xrow[(long long)ix * ldx] : 0.f;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bv[s], acc[i], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid][(i * 32 + 4 * hh + (r & 3) + 8 * (r >> 2)) * 32 + l31] = acc[i][r];
  __syncthreads();
  float* o = part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (MI * 1024);
  for (int e = threadIdx.x; e < MI * 1024; e += 256) o[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

__global__ __launch_bounds__(256) void conv_wgrad_stage2(const float* __restrict__ part, int njobs, int nchunks, int tile_m, int Cin,
                                                         int Cout, int taps, int n_ci_t, float* __restrict__ dw, int accumulate) {
  const long long total = (long long)Cout * taps * Cin;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int ci = (int)(e % Cin); long long t = e / Cin;
  const int tap = (int)(t % taps); const int co = (int)(t / taps);
  const int co_t = co / tile_m, m = co % tile_m, ci_t = ci / 32, n = ci % 32;
  const int job = (co_t * n_ci_t + ci_t) * taps + tap;
  const float* p = part + (long long)job * (tile_m * 32) + m * 32 + n;
  const long long stride = (long long)njobs * (tile_m * 32);
  float s0 = 0.f, s1 = 0.f;
  int k = 0;
  for (; k + 1 < nchunks; k += 2) { s0 += p[k * stride]; s1 += p[(k + 1) * stride]; }
  if (k < nchunks) s0 += p[k * stride];
  const float v = s0 + s1;
  dw[e] = accumulate ? dw[e] + v : v;
}

static void conv_wgrad_plan(int B, int H, int Cin, int Cout, int KH, int KW, int& MI, int& n_co_t, int& n_ci_t, int& njobs,
                            int& nchunks, int& rpc) {
  MI = Cout > 32 ? 2 : 1;
  n_co_t = (Cout + 32 * MI - 1) / (32 * MI);
  n_ci_t = (Cin + 31) / 32;
  njobs = n_co_t * n_ci_t * KH * KW;
  const long long nrows = (long long)B * H;
  long long want = (2048 + njobs - 1) / njobs;                 // ~8 workgroups per CU in total
  if (want > (nrows + 3) / 4) want = (nrows + 3) / 4;          // at least 4 image rows (one per wave) per chunk
  if (want < 1) want = 1;
  rpc = (int)((nrows + want - 1) / want);
  nchunks = (int)((nrows + rpc - 1) / rpc);
}

// a 1x1 convolution / linear layer sees only a list of pixels: cut it into rows of 8..512 pixels so that the row walk above has
// many rows to spread over chunks and waves (a GEMM arrives as B = 1, H = 1, W = M)
static void conv_wgrad_refactor(int& B, int& H, int& W, int KH, int KW) {
  if (KH != 1 || KW != 1) return;
  const long long M = (long long)B * H * W;
  int w = 0;
  for (int cand = 512; cand >= 8; cand >>= 1)
    if (M % cand == 0) { w = cand; break; }
  if (!w || M / w > 0x7fffffffLL) return;
  B = 1; H = (int)(M / w); W = w;
}

extern "C" long long ff_conv2d_wgrad_workspace(int B, int H, int W, int Cin, int Cout, int KH, int KW) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return -1;
  conv_wgrad_refactor(B, H, W, KH, KW);
  int MI, a, b, njobs, nchunks, rpc;
  conv_wgrad_plan(B, H, Cin, Cout, KH, KW, MI, a, b, njobs, nchunks, rpc);
  return (long long)njobs * nchunks * MI * 1024;
}

extern "C" int ff_conv2d_wgrad(const float* in, int ldx, const float* dz, int ldz, float* dw, int B, int H, int W, int Cin,
                               int Cout, int KH, int KW, int py, int px, int accumulate, float* work, long long work_floats,
                               void* stream) {
  FF_CHECK_ARG(in && dz && dw && work && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "ff_conv2d_wgrad: bad args");
  FF_CHECK_ARG(ldx >= Cin && ldz >= Cout, "ff_conv2d_wgrad: row strides smaller than the channel counts");
  FF_CHECK_ARG(KH >= 1 && KW >= 1 && 2 * py == KH - 1 && 2 * px == KW - 1, "ff_conv2d_wgrad: stride-1 'same' convolutions only (KH = 2 py + 1)");
  conv_wgrad_refactor(B, H, W, KH, KW);
  int MI, n_co_t, n_ci_t, njobs, nchunks, rpc;
  conv_wgrad_plan(B, H, Cin, Cout, KH, KW, MI, n_co_t, n_ci_t, njobs, nchunks, rpc);
  FF_CHECK_ARG(work_floats >= (long long)njobs * nchunks * MI * 1024, "ff_conv2d_wgrad: workspace too small (need %lld floats)",
               (long long)njobs * nchunks * MI * 1024);
  FF_CHECK_ARG(njobs <= 65535 * 16 && nchunks <= 65535, "ff_conv2d_wgrad: grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (MI == 2)
    hipLaunchKernelGGL(conv_wgrad_stage1<2>, dim3(njobs, nchunks), dim3(256), 0, st, in, ldx, dz, ldz, B, H, W, Cin, Cout, KH, KW, py, px,
                       n_ci_t, rpc, work);
  else
    hipLaunchKernelGGL(conv_wgrad_stage1<1>, dim3(njobs, nchunks), dim3(256), 0, st, in, ldx, dz, ldz, B, H, W, Cin, Cout, KH, KW, py, px,
                       n_ci_t, rpc, work);
  const long long total = (long long)Cout * KH * KW * Cin;
  hipLaunchKernelGGL(conv_wgrad_stage2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, work, njobs, nchunks, 32 * MI, Cin, Cout,
                     KH * KW, n_ci_t, dw, accumulate);
  FF_LAUNCH_CHECK("ff_conv2d_wgrad");
  return FF_OK;
}

// wt[ci][(KH-1-ky)*KW + (KW-1-kx)][co] = w[co][ky*KW + kx][ci]: conv(dz, wt) with the same padding is the data gradient
__global__ __launch_bounds__(256) void conv_weight_flipT_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int taps) {
  const long long total = (long long)Cout * taps * Cin;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int co = (int)(e % Cout); long long t = e / Cout;
    const int tp = (int)(t % taps); const int ci = (int)(t / taps);
    wt[e] = w[((long long)co * taps + (taps - 1 - tp)) * Cin + ci];
  }
}

extern "C" int ff_conv_weight_flipT(const float* w, float* wt, int Cout, int Cin, int KH, int KW, void* stream) {
  FF_CHECK_ARG(w && wt && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "ff_conv_weight_flipT: bad args");
  long long nb = ((long long)Cout * Cin * KH * KW + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(conv_weight_flipT_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, wt, Cout, Cin, KH * KW);
  FF_LAUNCH_CHECK("ff_conv_weight_flipT");
  return FF_OK;
}

// depth-wise: reversed taps (data gradient = depth-wise conv with them) and the weight gradient
__global__ __launch_bounds__(256) void taps_reverse_kernel(const float* __restrict__ w, float* __restrict__ o, int taps, int C) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < taps * C; e += gridDim.x * 256) o[e] = w[(taps - 1 - e / C) * C + e % C];
}
extern "C" int ff_taps_reverse(const float* w, float* out, int taps, int C, void* stream) {
  FF_CHECK_ARG(w && out && taps > 0 && C > 0, "ff_taps_reverse: bad args");
  hipLaunchKernelGGL(taps_reverse_kernel, dim3((taps * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, out, taps, C);
  FF_LAUNCH_CHECK("ff_taps_reverse");
  return FF_OK;
}

// dw[tap][c] = sum_{b,y,x} dy[b,y,x,c] * in[b, y+ky-py, x+kx-px, c]; lane = channel, waves = row phases, taps in registers
template <int KH, int KW>
__global__ __launch_bounds__(256) void dwconv_wgrad_stage1(const float* __restrict__ in, int ldx, const float* __restrict__ dy, int ldy,
                                                           int B, int H, int W, int C, int rows_per_chunk, float* __restrict__ part) {
  constexpr int T = KH * KW, py = KH / 2, px = KW / 2;
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const bool cok = c < C;
  float acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.f;
  const long long nrows = (long long)B * H;
  const long long row0 = (long long)blockIdx.y * rows_per_chunk;
  long long row1 = row0 + rows_per_chunk;
  if (row1 > nrows) row1 = nrows;
  if (cok) {
    for (long long row = row0 + wid; row < row1; row += 4) {
      const int y = (int)(row % H);
      const long long b = row / H;
      const float* grow = dy + row * W * (long long)ldy + c;
      for (int x = 0; x < W; ++x) {
        const float g = grow[(long long)x * ldy];
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
          const int iy = y + ky - py;
          if ((unsigned)iy >= (unsigned)H) continue;
          const float* xr = in + ((b * H + iy) * (long long)W) * ldx + c;
#pragma unroll
          for (int kx = 0; kx < KW; ++kx) {
            const int ix = x + kx - px;
            if ((unsigned)ix < (unsigned)W) acc[ky * KW + kx] += g * xr[(long long)ix * ldx];
          }
        }
      }
    }
  }
  float* o = part + (long long)blockIdx.y * T * C;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    red[wid][lane] = acc[t];
    __syncthreads();
    if (wid == 0 && cok) o[t * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    __syncthreads();
  }
}

static void dw_wgrad_plan(int B, int H, int C, int& nchunks, int& rpc) {
  const long long nrows = (long long)B * H;
  const int ct = (C + 63) / 64;
  long long want = (2048 + ct - 1) / ct;
  if (want > (nrows + 3) / 4) want = (nrows + 3) / 4;
  if (want < 1) want = 1;
  rpc = (int)((nrows + want - 1) / want);
  nchunks = (int)((nrows + rpc - 1) / rpc);
}

extern "C" long long ff_dwconv2d_wgrad_workspace(int B, int H, int W, int C, int KH, int KW) {
  if (B <= 0 || H <= 0 || C <= 0) return -1;
  int nchunks, rpc;
  dw_wgrad_plan(B, H, C, nchunks, rpc);
  return (long long)nchunks * KH * KW * C;
}

extern "C" int ff_dwconv2d_wgrad(const float* in, int ldx, const float* dy, int ldy, float* dw, int B, int H, int W, int C, int KH,
                                 int KW, int accumulate, float* work, long long work_floats, void* stream) {
  FF_CHECK_ARG(in && dy && dw && work && B > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && ldy >= C, "ff_dwconv2d_wgrad: bad args");
  int nchunks, rpc;
  dw_wgrad_plan(B, H, C, nchunks, rpc);
  const int T = KH * KW;
  FF_CHECK_ARG(work_floats >= (long long)nchunks * T * C, "ff_dwconv2d_wgrad: workspace too small (need %lld floats)", (long long)nchunks * T * C);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((C + 63) / 64, nchunks), block(256);
#define DW_CASE(kh, kw) hipLaunchKernelGGL((dwconv_wgrad_stage1<kh, kw>), grid, block, 0, st, in, ldx, dy, ldy, B, H, W, C, rpc, work)
  if (KH == 3 && KW == 3) DW_CASE(3, 3);
  else if (KH == 5 && KW == 5) DW_CASE(5, 5);
  else if (KH == 1 && KW == 21) DW_CASE(1, 21);
  else if (KH == 21 && KW == 1) DW_CASE(21, 1);
  else { ff_set_error("ff_dwconv2d_wgrad: built for 3x3, 5x5, 1x21 and 21x1 (the LKA chain), got %dx%d", KH, KW); return FF_ERR_ARG; }
#undef DW_CASE
  // stage 2: [nchunks][T*C] -> [T*C]  (one "group", T*C columns)
  launch_reduce_stage2(work, nchunks, T * C, 1, 1.0f, dw, accumulate, st);
  FF_LAUNCH_CHECK("ff_dwconv2d_wgrad");
  return FF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Adjoint of bilinear F.interpolate(align_corners=False) in gather form (deterministic): every INPUT pixel sums the output
// pixels whose two taps per axis touch it.  The candidate range is widened by one on each side and every candidate is checked with
// exactly the forward's index arithmetic (csrc/resample.hip), so the weights are the forward's bit for bit.
__device__ __forceinline__ void bl_fwd(int o, int n_in, int n_out, float sc, int& i0, int& i1, float& l) {
  if (n_in == n_out) { i0 = i1 = o; l = 0.f; return; }
  const float s = fmaxf(sc * ((float)o + 0.5f) - 0.5f, 0.f);
  i0 = min((int)floorf(s), n_in - 1); i1 = min(i0 + 1, n_in - 1);
  l = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
}
__device__ __forceinline__ void bl_range(int i, int n_in, int n_out, float sc, int& lo, int& hi) {
  if (n_in == n_out) { lo = hi = i; return; }
  lo = (int)floorf(((float)i - 1.0f + 0.5f) / sc - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.0f + 0.5f) / sc - 0.5f) + 1;
  if (i == n_in - 1) hi = n_out - 1;                               // clamped taps at the far border
  if (lo < 0) lo = 0;
  if (hi > n_out - 1) hi = n_out - 1;
}

__global__ __launch_bounds__(256) void resize_bilinear_adj_kernel(const float* __restrict__ dy, int ldy, int Ho, int Wo,
                                                                  float* __restrict__ dx, int ldx, int Hi, int Wi, int B, int C,
                                                                  float sh, float sw, float mul, int accumulate) {
  const long long total = (long long)B * Hi * Wi * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C;
    const int x = (int)(t % Wi); t /= Wi;
    const int y = (int)(t % Hi); const int b = (int)(t / Hi);
    int ylo, yhi, xlo, xhi;
    bl_range(y, Hi, Ho, sh, ylo, yhi);
    bl_range(x, Wi, Wo, sw, xlo, xhi);
    float s = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      int y0, y1; float ly;
      bl_fwd(oy, Hi, Ho, sh, y0, y1, ly);
      const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      const float* grow = dy + (((long long)b * Ho + oy) * Wo) * ldy + c;
      float rs = 0.f;
      for (int ox = xlo; ox <= xhi; ++ox) {
        int x0, x1; float lx;
        bl_fwd(ox, Wi, Wo, sw, x0, x1, lx);
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        if (wx != 0.f) rs += wx * grow[(long long)ox * ldy];
      }
      s += wy * rs;
    }
    float* o = dx + (((long long)b * Hi + y) * Wi + x) * ldx + c;
    s *= mul;
    *o = accumulate ? *o + s : s;
  }
}

extern "C" int ff_resize_bilinear_adj(const float* dy, int ldy, int Ho, int Wo, float* dx, int ldx, int Hi, int Wi, int B, int C,
                                      float scale_h, float scale_w, float mul, int accumulate, void* stream) {
  FF_CHECK_ARG(dy && dx && B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && ldy >= C && ldx >= C, "ff_resize_bilinear_adj: bad args");
  FF_CHECK_ARG(scale_h > 0.f && scale_w > 0.f, "ff_resize_bilinear_adj: scales must be positive");
  long long nb = ((long long)B * Hi * Wi * C + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(resize_bilinear_adj_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dy, ldy, Ho, Wo, dx, ldx, Hi, Wi,
                     B, C, scale_h, scale_w, mul, accumulate);
  FF_LAUNCH_CHECK("ff_resize_bilinear_adj");
  return FF_OK;
}

// adjoint of avg_pool2d(2): dx[b, y, x, c] = 0.25 * dy[b, y/2, x/2, c] for y < 2*(H/2), x < 2*(W/2); zero on an odd last row/column
__global__ __launch_bounds__(256) void avgpool2_adj_kernel(const float* __restrict__ dy, int ldy, float* __restrict__ dx, int ldx, int B,
                                                           int H, int W, int C, int accumulate) {
  const int Ho = H / 2, Wo = W / 2;
  const long long total = (long long)B * H * W * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); const int b = (int)(t / H);
    float v = 0.f;
    if (y / 2 < Ho && x / 2 < Wo) v = 0.25f * dy[(((long long)b * Ho + y / 2) * Wo + x / 2) * ldy + c];
    float* o = dx + (((long long)b * H + y) * W + x) * ldx + c;
    *o = accumulate ? *o + v : v;
  }
}
extern "C" int ff_avgpool2_adj(const float* dy, int ldy, float* dx, int ldx, int B, int H, int W, int C, int accumulate, void* stream) {
  FF_CHECK_ARG(dy && dx && B > 0 && H >= 2 && W >= 2 && C > 0 && ldy >= C && ldx >= C, "ff_avgpool2_adj: bad args");
  long long nb = ((long long)B * H * W * C + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(avgpool2_adj_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dy, ldy, dx, ldx, B, H, W, C, accumulate);
  FF_LAUNCH_CHECK("ff_avgpool2_adj");
  return FF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm backward, one wave per row (C <= 256):
//   xh = (x - mean) rstd;  gh = dy * gamma;  dx = rstd * (gh - mean_c(gh) - xh * mean_c(gh * xh))
// dgamma / dbeta: per-workgroup partial column sums over its rows -> stage 2.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int ldy,
                                                            const float* __restrict__ gamma, float eps, float* __restrict__ dx,
                                                            int lddx, long long rows, int C, int rows_per_blk,
                                                            float* __restrict__ part) {
  __shared__ float sg[4][256], sb[4][256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  const long long r0 = (long long)blockIdx.x * rows_per_blk;
  long long r1 = r0 + rows_per_blk;
  if (r1 > rows) r1 = rows;
  for (long long r = r0 + wid; r < r1; r += 4) {
    float xv[4], gv[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      xv[i] = c < C ? x[r * ldx + c] : 0.f;
      gv[i] = c < C ? dy[r * ldy + c] : 0.f;
      s += xv[i];
    }
    const float mean = tw_wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float d = (lane + 64 * i < C) ? xv[i] - mean : 0.f; q += d * d; }
    const float rstd = 1.0f / sqrtf(tw_wave_sum(q) / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < C) {
        const float xh = (xv[i] - mean) * rstd;
        const float gh = gv[i] * gamma[c];
        ag[i] += gv[i] * xh;
        ab[i] += gv[i];
        s1 += gh; s2 += gh * xh;
        xv[i] = xh; gv[i] = gh;
      }
    }
    const float m1 = tw_wave_sum(s1) / (float)C, m2 = tw_wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < C) dx[r * lddx + c] = rstd * (gv[i] - m1 - xv[i] * m2);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { sg[wid][lane + 64 * i] = ag[i]; sb[wid][lane + 64 * i] = ab[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    part[(long long)blockIdx.x * 2 * C + c] = (sg[0][c] + sg[1][c]) + (sg[2][c] + sg[3][c]);
    part[(long long)blockIdx.x * 2 * C + C + c] = (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]);
  }
}

static void ln_bwd_plan(long long rows, int& nblk, int& rpb) {
  long long want = (rows + 63) / 64;
  if (want > 2048) want = 2048;
  if (want < 1) want = 1;
  rpb = (int)((rows + want - 1) / want);
  nblk = (int)((rows + rpb - 1) / rpb);
}
extern "C" long long ff_layernorm_bwd_workspace(long long rows, int C) {
  if (rows <= 0 || C <= 0) return -1;
  int nblk, rpb;
  ln_bwd_plan(rows, nblk, rpb);
  return (long long)nblk * 2 * C;
}
extern "C" int ff_layernorm_bwd(const float* x, int ldx, const float* dy, int ldy, const float* gamma, float eps, float* dx, int lddx,
                                long long rows, int C, float* dgamma_dbeta, int accumulate, float* work, long long work_floats,
                                void* stream) {
  FF_CHECK_ARG(x && dy && gamma && dx && dgamma_dbeta && work && rows > 0 && C > 0 && C <= 256, "ff_layernorm_bwd: bad args (C <= 256)");
  FF_CHECK_ARG(ldx >= C && ldy >= C && lddx >= C, "ff_layernorm_bwd: row strides smaller than C");
  int nblk, rpb;
  ln_bwd_plan(rows, nblk, rpb);
  FF_CHECK_ARG(work_floats >= (long long)nblk * 2 * C, "ff_layernorm_bwd: workspace too small (need %lld floats)", (long long)nblk * 2 * C);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblk), dim3(256), 0, st, x, ldx, dy, ldy, gamma, eps, dx, lddx, rows, C, rpb, work);
  launch_reduce_stage2(work, nblk, 2 * C, 1, 1.0f, dgamma_dbeta, accumulate, st);
  FF_LAUNCH_CHECK("ff_layernorm_bwd");
  return FF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// BatchNorm2d, training mode.  Rows = pixels of G image groups of rpg rows each (one nn.BatchNorm2d CALL per group: the LKA block is
// applied band by band / expert by expert, large_kernel_attention.py:236-240,406-411, each call with its own batch statistics).
//   ff_bn_train_finish: from sum(x) and sum(x^2) (or, centered != 0, sum((x - mean)^2)) per (group, channel): mean, biased var -> scale = gamma rstd, shift = beta - mean scale
//   (apply with ff_ew_fma kind 2), and the running statistics updated call by call: r = (1-m) r + m stat (unbiased variance).
__global__ __launch_bounds__(256) void bn_train_finish_kernel(const float* __restrict__ s1, const float* __restrict__ s2, int centered, int G, int C,
                                                              long long count, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, float momentum,
                                                              float* __restrict__ rmean, float* __restrict__ rvar,
                                                              float* __restrict__ mean_rstd, float* __restrict__ scale,
                                                              float* __restrict__ shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float n = (float)count;
  float rm = rmean ? rmean[c] : 0.f, rv = rvar ? rvar[c] : 0.f;
  for (int g = 0; g < G; ++g) {
    const float mean = s1[g * C + c] / n;
    float var = centered ? s2[g * C + c] / n : s2[g * C + c] / n - mean * mean;   // biased; centered: s2 = sum (x - mean)^2
    if (var < 0.f) var = 0.f;
    const float rstd = 1.0f / sqrtf(var + eps);
    mean_rstd[(g * C + c) * 2] = mean;
    mean_rstd[(g * C + c) * 2 + 1] = rstd;
    const float sc = gamma[c] * rstd;
    scale[g * C + c] = sc;
    shift[g * C + c] = beta[c] - mean * sc;
    rm = (1.f - momentum) * rm + momentum * mean;
    rv = (1.f - momentum) * rv + momentum * (count > 1 ? var * n / (n - 1.f) : var);
  }
  if (rmean) rmean[c] = rm;
  if (rvar) rvar[c] = rv;
}

extern "C" int ff_bn_train_finish(const float* sum_x, const float* sum_x2, int centered, int G, int C, long long count, const float* gamma,
                                  const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                  float* mean_rstd, float* scale, float* shift, void* stream) {
  FF_CHECK_ARG(sum_x && sum_x2 && gamma && beta && mean_rstd && scale && shift && G > 0 && C > 0 && count > 0, "ff_bn_train_finish: bad args");
  hipLaunchKernelGGL(bn_train_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sum_x, sum_x2, centered, G, C, count, gamma,
                     beta, eps, momentum, running_mean, running_var, mean_rstd, scale, shift);
  FF_LAUNCH_CHECK("ff_bn_train_finish");
  return FF_OK;
}

// backward apply:  dx = gamma rstd (dy - sum(dy)/n - xh * sum(dy xh)/n),  xh = (x - mean) rstd ; sdy / sdyxh are [G][C]
__global__ __launch_bounds__(256) void bn_train_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int ldy,
                                                           const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ sdy, const float* __restrict__ sdyxh,
                                                           float* __restrict__ dx, int lddx, long long rows, int C, long long rpg) {
  const long long total = rows * C;
  const float inv = 1.0f / (float)rpg;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C;
    const int c = (int)(i - r * C);
    const long long gc = (r / rpg) * C + c;
    const float mean = mean_rstd[gc * 2], rstd = mean_rstd[gc * 2 + 1];
    const float xh = (x[r * ldx + c] - mean) * rstd;
    dx[r * lddx + c] = gamma[c] * rstd * (dy[r * ldy + c] - sdy[gc] * inv - xh * sdyxh[gc] * inv);
  }
}

// xhat = (x - mean) * rstd  (needed for dgamma = sum dy * xhat)
__global__ __launch_bounds__(256) void bn_xhat_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mean_rstd,
                                                      float* __restrict__ out, int ldo, long long rows, int C, long long rpg) {
  const long long total = rows * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C;
    const int c = (int)(i - r * C);
    const long long gc = (r / rpg) * C + c;
    out[r * ldo + c] = (x[r * ldx + c] - mean_rstd[gc * 2]) * mean_rstd[gc * 2 + 1];
  }
}
extern "C" int ff_bn_xhat(const float* x, int ldx, const float* mean_rstd, float* out, int ldo, long long rows, int C,
                          long long rows_per_group, void* stream) {
  FF_CHECK_ARG(x && mean_rstd && out && rows > 0 && C > 0 && rows_per_group > 0 && rows % rows_per_group == 0 && ldx >= C && ldo >= C, "ff_bn_xhat: bad args");
  long long nb = (rows * C + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(bn_xhat_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ldx, mean_rstd, out, ldo, rows, C, rows_per_group);
  FF_LAUNCH_CHECK("ff_bn_xhat");
  return FF_OK;
}
extern "C" int ff_bn_train_bwd(const float* x, int ldx, const float* dy, int ldy, const float* mean_rstd, const float* gamma,
                               const float* sum_dy, const float* sum_dy_xhat, float* dx, int lddx, long long rows, int C,
                               long long rows_per_group, void* stream) {
  FF_CHECK_ARG(x && dy && mean_rstd && gamma && sum_dy && sum_dy_xhat && dx && rows > 0 && C > 0, "ff_bn_train_bwd: bad args");
  FF_CHECK_ARG(rows_per_group > 0 && rows % rows_per_group == 0 && ldx >= C && ldy >= C && lddx >= C, "ff_bn_train_bwd: bad strides / groups");
  long long nb = (rows * C + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(bn_train_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, ldy, mean_rstd, gamma, sum_dy,
                     sum_dy_xhat, dx, lddx, rows, C, rows_per_group);
  FF_LAUNCH_CHECK("ff_bn_train_bwd");
  return FF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Per-pixel attention core over NT tokens (9 bands / 3 experts), `heads` heads of d = 16, with dropout on the attention weights.
// qkv rows [(p*NT + i)][3E] = q | k | v, E = heads*16; out rows [(p*NT + i)][E].  One thread per (pixel, query token, head).
// Dropout mask: keep = hash(seed, p, head, i, j) >= drop_p, weights scaled by 1/(1-p) (nn.MultiheadAttention applies F.dropout to the
// softmax output); the same hash regenerates the mask in the backward pass.
__device__ __forceinline__ float mha_uniform(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;        // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

template <int NT>
__global__ __launch_bounds__(256) void band_mha_train_kernel(const float* __restrict__ qkv, float* __restrict__ out, long long P,
                                                             int heads, float drop_p, unsigned long long seed) {
  const int d = 16, E = heads * d;
  const long long total = P * NT * heads;
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int h = (int)(idx % heads);
    long long t = idx / heads;
    const int i = (int)(t % NT);
    const long long p = t / NT;
    const float* base = qkv + p * NT * 3 * E;
    f32x4 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const f32x4*>(base + (long long)i * 3 * E + h * d + 4 * u) * 0.25f;
    float s[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float a = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 k = *reinterpret_cast<const f32x4*>(base + (long long)j * 3 * E + E + h * d + 4 * u);
        a += q[u][0] * k[0] + q[u][1] * k[1] + q[u][2] * k[2] + q[u][3] * k[3];
      }
      s[j] = a;
      mx = fmaxf(mx, a);
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
    f32x4 o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float w = s[j] / den;
      if (drop_p > 0.f) w = mha_uniform(seed, ((unsigned long long)(p * heads + h) * NT + i) * NT + j) >= drop_p ? w * keep_scale : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] += *reinterpret_cast<const f32x4*>(base + (long long)j * 3 * E + 2 * E + h * d + 4 * u) * w;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(out + (p * NT + i) * E + h * d + 4 * u) = o[u];
  }
}

extern "C" int ff_band_mha_train(const float* qkv, float* out, long long P, int ntok, int heads, float drop_p,
                                 unsigned long long seed, void* stream) {
  FF_CHECK_ARG(qkv && out && P > 0 && heads > 0 && drop_p >= 0.f && drop_p < 1.f, "ff_band_mha_train: bad args");
  FF_CHECK_ARG(ntok == 9 || ntok == 3, "ff_band_mha_train: built for 9 tokens (frequency bands) or 3 (experts), got %d", ntok);
  long long nb = (P * ntok * heads + 255) / 256;
  if (nb > 16384) nb = 16384;
  if (ntok == 9) hipLaunchKernelGGL(band_mha_train_kernel<9>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, qkv, out, P, heads, drop_p, seed);
  else hipLaunchKernelGGL(band_mha_train_kernel<3>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, qkv, out, P, heads, drop_p, seed);
  FF_LAUNCH_CHECK("ff_band_mha_train");
  return FF_OK;
}

// Backward in two passes over a scratch tensor T [P][heads][NT][NT][2] = (dS_ij, W_ij):
//   pass A, one thread per (pixel, query i, head): recompute P_i and W_i = dropout(P_i); dW_ij = dO_i . v_j; dP = mask/(1-p) dW;
//           dS_ij = 0.25 P_ij (dP_ij - sum_j dP_ij P_ij);  dq_i = sum_j dS_ij k_j
//   pass B, one thread per (pixel, key j, head): dk_j = sum_i dS_ij q_i;  dv_j = sum_i W_ij dO_i
template <int NT>
__global__ __launch_bounds__(256) void band_mha_bwd_a_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                             float* __restrict__ dqkv, float* __restrict__ T, long long P, int heads,
                                                             float drop_p, unsigned long long seed) {
  constexpr int d = 16;
  const int E = heads * d;
  const long long total = P * NT * heads;
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int h = (int)(idx % heads);
    long long t = idx / heads;
    const int i = (int)(t % NT);
    const long long p = t / NT;
    const float* base = qkv + p * NT * 3 * E;
    f32x4 q[4], go[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      q[u] = *reinterpret_cast<const f32x4*>(base + (long long)i * 3 * E + h * d + 4 * u) * 0.25f;
      go[u] = *reinterpret_cast<const f32x4*>(dout + (p * NT + i) * E + h * d + 4 * u);
    }
    float s[NT], dp[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 k = *reinterpret_cast<const f32x4*>(base + (long long)j * 3 * E + E + h * d + 4 * u);
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)j * 3 * E + 2 * E + h * d + 4 * u);
        a += q[u][0] * k[0] + q[u][1] * k[1] + q[u][2] * k[2] + q[u][3] * k[3];
        b += go[u][0] * v[0] + go[u][1] * v[1] + go[u][2] * v[2] + go[u][3] * v[3];
      }
      s[j] = a; dp[j] = b;
      mx = fmaxf(mx, a);
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
    float dot = 0.f;
    float* trow = T + (((p * heads + h) * NT + i) * NT) * 2;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      s[j] /= den;
      float m = 1.f;
      if (drop_p > 0.f) m = mha_uniform(seed, ((unsigned long long)(p * heads + h) * NT + i) * NT + j) >= drop_p ? keep_scale : 0.f;
      trow[2 * j + 1] = s[j] * m;                                    // W_ij
      dp[j] *= m;
      dot += dp[j] * s[j];
    }
    f32x4 dq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) dq[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float ds = s[j] * (dp[j] - dot) * 0.25f;
      trow[2 * j] = ds;
#pragma unroll
      for (int u = 0; u < 4; ++u) dq[u] += *reinterpret_cast<const f32x4*>(base + (long long)j * 3 * E + E + h * d + 4 * u) * ds;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(dqkv + (p * NT + i) * 3 * E + h * d + 4 * u) = dq[u];
  }
}

template <int NT>
__global__ __launch_bounds__(256) void band_mha_bwd_b_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                             float* __restrict__ dqkv, const float* __restrict__ T, long long P,
                                                             int heads) {
  constexpr int d = 16;
  const int E = heads * d;
  const long long total = P * NT * heads;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int h = (int)(idx % heads);
    long long t = idx / heads;
    const int j = (int)(t % NT);
    const long long p = t / NT;
    const float* base = qkv + p * NT * 3 * E;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { dk[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[u] = dk[u]; }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const float* tp = T + ((((p * heads + h) * NT + i) * NT) + j) * 2;
      const float ds = tp[0], w = tp[1];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        dk[u] += *reinterpret_cast<const f32x4*>(base + (long long)i * 3 * E + h * d + 4 * u) * ds;          // q unscaled: the 0.25 is in dS
        dv[u] += *reinterpret_cast<const f32x4*>(dout + (p * NT + i) * E + h * d + 4 * u) * w;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      *reinterpret_cast<f32x4*>(dqkv + (p * NT + j) * 3 * E + E + h * d + 4 * u) = dk[u];
      *reinterpret_cast<f32x4*>(dqkv + (p * NT + j) * 3 * E + 2 * E + h * d + 4 * u) = dv[u];
    }
  }
}

extern "C" long long ff_band_mha_bwd_workspace(long long P, int ntok, int heads) {
  return P > 0 && ntok > 0 && heads > 0 ? P * heads * ntok * ntok * 2 : -1;
}

extern "C" int ff_band_mha_bwd(const float* qkv, const float* dout, float* dqkv, long long P, int ntok, int heads, float drop_p,
                               unsigned long long seed, float* work, long long work_floats, void* stream) {
  FF_CHECK_ARG(qkv && dout && dqkv && work && P > 0 && heads > 0 && drop_p >= 0.f && drop_p < 1.f, "ff_band_mha_bwd: bad args");
  FF_CHECK_ARG(ntok == 9 || ntok == 3, "ff_band_mha_bwd: built for 9 or 3 tokens, got %d", ntok);
  FF_CHECK_ARG(work_floats >= P * heads * ntok * ntok * 2, "ff_band_mha_bwd: workspace too small (need %lld floats)", P * heads * ntok * ntok * 2);
  long long nb = (P * ntok * heads + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipStream_t st = (hipStream_t)stream;
  if (ntok == 9) {
    hipLaunchKernelGGL(band_mha_bwd_a_kernel<9>, dim3((unsigned)nb), dim3(256), 0, st, qkv, dout, dqkv, work, P, heads, drop_p, seed);
    hipLaunchKernelGGL(band_mha_bwd_b_kernel<9>, dim3((unsigned)nb), dim3(256), 0, st, qkv, dout, dqkv, work, P, heads);
  } else {
    hipLaunchKernelGGL(band_mha_bwd_a_kernel<3>, dim3((unsigned)nb), dim3(256), 0, st, qkv, dout, dqkv, work, P, heads, drop_p, seed);
    hipLaunchKernelGGL(band_mha_bwd_b_kernel<3>, dim3((unsigned)nb), dim3(256), 0, st, qkv, dout, dqkv, work, P, heads);
  }
  FF_LAUNCH_CHECK("ff_band_mha_bwd");
  return FF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// dynamic_gates backward (fusion_network.py:226-234): g = sigmoid(10 (graw - th)), th = 0.7 - 0.4 dif,
// gates = max(g, [g >= 0.99 max g] * 0.9): gradient flows to g where g > floor (half of it where equal, as torch.maximum does).
__global__ __launch_bounds__(256) void dynamic_gates_bwd_kernel(const float* __restrict__ graw, const float* __restrict__ dif,
                                                                const float* __restrict__ dgates, float* __restrict__ dgraw,
                                                                float* __restrict__ ddif, long long P) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
    const float th = 0.7f - 0.4f * dif[p];
    float g[3], mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 3; ++e) { g[e] = 1.0f / (1.0f + expf(-10.0f * (graw[p * 3 + e] - th))); mx = fmaxf(mx, g[e]); }
    float dth = 0.f;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const float fl = (g[e] >= mx * 0.99f) ? 0.9f : 0.f;
      const float pass = g[e] > fl ? 1.f : (g[e] == fl ? 0.5f : 0.f);
      const float dz = dgates[p * 3 + e] * pass * g[e] * (1.f - g[e]) * 10.f;
      dgraw[p * 3 + e] = dz;
      dth -= dz;
    }
    ddif[p] = -0.4f * dth;
  }
}
extern "C" int ff_dynamic_gates_bwd(const float* graw, const float* dif, const float* dgates, float* dgraw, float* ddif, long long P,
                                    void* stream) {
  FF_CHECK_ARG(graw && dif && dgates && dgraw && ddif && P > 0, "ff_dynamic_gates_bwd: bad args");
  long long nb = (P + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(dynamic_gates_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, graw, dif, dgates, dgraw, ddif, P);
  FF_LAUNCH_CHECK("ff_dynamic_gates_bwd");
  return FF_OK;
}

// [A][B][C] -> [B][A][C]
__global__ __launch_bounds__(256) void permute_rows_kernel(const float* __restrict__ in, float* __restrict__ out, long long A, long long Bd, int C) {
  const long long total = A * Bd * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C;
    const long long a = t % A; const long long b = t / A;            // output index (b, a, c)
    out[i] = in[(a * Bd + b) * C + c];
  }
}
extern "C" int ff_permute_rows(const float* in, float* out, long long A, long long Bd, int C, void* stream) {
  FF_CHECK_ARG(in && out && A > 0 && Bd > 0 && C > 0, "ff_permute_rows: bad args");
  long long nb = (A * Bd * C + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, out, A, Bd, C);
  FF_LAUNCH_CHECK("ff_permute_rows");
  return FF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Half-spectrum helpers of the learnable FFT mask (multi_domain_frequency.py:362-385).  Spectra are [planes][H][Wf][2] (re, im).
//   ff_spec_mask_mul : Y = X * m[ky][kx]                           (m real [H][Wf], shared by all planes)
//   ff_spec_mask_grad: dm[ky][kx] = sum_planes Re(gY conj X) * (2 for the columns that have a conjugate partner, else 1)
//                      where gY = rfft2(g): torch's c2r backward doubles exactly those columns.
__global__ __launch_bounds__(256) void spec_mask_mul_kernel(const float* __restrict__ X, const float* __restrict__ m, float* __restrict__ Y,
                                                            int planes, int HWf) {
  const long long total = (long long)planes * HWf;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const float mv = m[i % HWf];
    Y[2 * i] = X[2 * i] * mv; Y[2 * i + 1] = X[2 * i + 1] * mv;
  }
}
extern "C" int ff_spec_mask_mul(const float* X, const float* m, float* Y, int planes, int H, int Wf, void* stream) {
  FF_CHECK_ARG(X && m && Y && planes > 0 && H > 0 && Wf > 0, "ff_spec_mask_mul: bad args");
  long long nb = ((long long)planes * H * Wf + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(spec_mask_mul_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, X, m, Y, planes, H * Wf);
  FF_LAUNCH_CHECK("ff_spec_mask_mul");
  return FF_OK;
}
__global__ __launch_bounds__(256) void spec_mask_grad_kernel(const float* __restrict__ gY, const float* __restrict__ X, float* __restrict__ dm,
                                                             int planes, int H, int Wf, int W, int accumulate) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= H * Wf) return;
  const int kx = e % Wf;
  const bool paired = kx >= 1 && kx <= (W - 1) / 2;
  float s = 0.f;
  for (int pl = 0; pl < planes; ++pl) {
    const long long o = ((long long)pl * H * Wf + e) * 2;
    s += gY[o] * X[o] + gY[o + 1] * X[o + 1];
  }
  s *= paired ? 2.f : 1.f;
  dm[e] = accumulate ? dm[e] + s : s;
}
extern "C" int ff_spec_mask_grad(const float* gY, const float* X, float* dm, int planes, int H, int W, int accumulate, void* stream) {
  FF_CHECK_ARG(gY && X && dm && planes > 0 && H > 0 && W > 1, "ff_spec_mask_grad: bad args");
  const int Wf = W / 2 + 1;
  hipLaunchKernelGGL(spec_mask_grad_kernel, dim3((H * Wf + 255) / 256), dim3(256), 0, (hipStream_t)stream, gY, X, dm, planes, H, Wf, W, accumulate);
  FF_LAUNCH_CHECK("ff_spec_mask_grad");
  return FF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Loss + optimizer over flat buffers.
//   ff_l1_loss_grad : loss = mean |clamp(sr,0,1) - hr| ; dsr = sign(clamp(sr) - hr) / n where 0 <= sr <= 1, else 0   (train.py:318-321)
//   ff_grad_sqnorm  : sum g^2 (clip_grad_norm_'s total norm, squared), fixed order
//   ff_adamw_ema_step : coef = min(1, max_norm / (sqrt(sqnorm) + 1e-6)); g *= coef; torch.optim.AdamW single-tensor update; EMA
__global__ __launch_bounds__(256) void l1_stage1_kernel(const float* __restrict__ sr, const float* __restrict__ hr, float* __restrict__ dsr,
                                                        long long n, float inv_n, float* __restrict__ part) {
  __shared__ float red[256];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float x = sr[i];
    const float d = fminf(fmaxf(x, 0.f), 1.f) - hr[i];
    s += fabsf(d);
    if (dsr) dsr[i] = (x >= 0.f && x <= 1.f) ? (d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f)) : 0.f;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) { if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void sum_finish_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) { if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}
extern "C" int ff_l1_loss_grad(const float* sr, const float* hr, float* dsr, long long n, float* loss, float* work, long long work_floats,
                               void* stream) {
  FF_CHECK_ARG(sr && hr && loss && work && n > 0 && work_floats >= 1024, "ff_l1_loss_grad: bad args (workspace: 1024 floats)");
  int nb = (int)((n + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_stage1_kernel, dim3(nb), dim3(256), 0, st, sr, hr, dsr, n, 1.0f / (float)n, work);
  hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, st, work, nb, 1.0f / (float)n, loss);
  FF_LAUNCH_CHECK("ff_l1_loss_grad");
  return FF_OK;
}

__global__ __launch_bounds__(256) void sqnorm_stage1_kernel(const float* __restrict__ g, long long n, float* __restrict__ part) {
  __shared__ float red[256];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += g[i] * g[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) { if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
extern "C" int ff_grad_sqnorm(const float* g, long long n, float* out_sqnorm, float* work, long long work_floats, void* stream) {
  FF_CHECK_ARG(g && out_sqnorm && work && n > 0 && work_floats >= 1024, "ff_grad_sqnorm: bad args (workspace: 1024 floats)");
  int nb = (int)((n + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_stage1_kernel, dim3(nb), dim3(256), 0, st, g, n, work);
  hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, st, work, nb, 1.0f, out_sqnorm);
  FF_LAUNCH_CHECK("ff_grad_sqnorm");
  return FF_OK;
}

// hyper [10] on the device: lr, beta1, beta2, eps, weight_decay, max_norm (<= 0: no clipping), ema_decay, step (1-based),
// step_size = lr / (1 - beta1^step), sqrt(1 - beta2^step) -- the last two computed by the host in double precision, as torch does
__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                        float* __restrict__ ema, long long n, const float* __restrict__ hyper,
                                                        const float* __restrict__ sqnorm) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], max_norm = hyper[5], decay = hyper[6];
  const float step_size = hyper[8], bc2s = hyper[9];
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (sqrtf(sqnorm[0]) + 1e-6f), 1.0f);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);                 // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;                // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(vi) / bc2s + eps;
    pi -= step_size * (mi / denom);
    g[i] = gi; m[i] = mi; v[i] = vi; p[i] = pi;
    if (ema) ema[i] = decay * ema[i] + (1.f - decay) * pi;
  }
}
extern "C" int ff_adamw_ema_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* ema, long long n,
                                 const float* hyper10, const float* sqnorm, void* stream) {
  FF_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && hyper10 && sqnorm && n > 0, "ff_adamw_ema_step: bad args");
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, ema, n, hyper10, sqnorm);
  FF_LAUNCH_CHECK("ff_adamw_ema_step");
  return FF_OK;
}

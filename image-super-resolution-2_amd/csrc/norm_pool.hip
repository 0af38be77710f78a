// Wave-reduction LayerNorm over the channel axis of NHWC / token tensors, global average pooling
// and the tiny per-image vector MLPs (squeeze-excite style gates).
//   ff_layernorm : nn.LayerNorm(C) on tokens (hat_arch.py:272,307,397,437; dat_arch.py:117,734-735)
//                  and NAFNet LayerNorm2d (nafnet_arch.py:35-41) -- identical in NHWC.
//   ff_pool_mean : AdaptiveAvgPool2d(1) (hat_arch.py:50; dat_arch.py:411,603; nafnet_arch.py:86)
//   ff_vec_mlp   : the 1x1-conv MLP that follows the pool (same lines), BN pre-folded by the host.
// All three are HBM-bound: one read of the row / tensor, one write.
#include "ff_common.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out,
                                                        int ldo, int rows, int C, const float* __restrict__ g,
                                                        const float* __restrict__ bta, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = in + (long long)row * ldi;
  float v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < C ? x[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    const float d = c < C ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  float* y = out + (long long)row * ldo;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < C) y[c] = (v[i] - mean) * rstd * g[c] + bta[c];
  }
}

extern "C" int ff_layernorm(const float* in, int ldi, float* out, int ldo, long long rows, int C, const float* gamma,
                            const float* beta, float eps, void* stream) {
  FF_CHECK_ARG(in && out && gamma && beta, "ff_layernorm: null pointer");
  FF_CHECK_ARG(C > 0 && C <= 1024 && ldi >= C && ldo >= C && rows > 0, "ff_layernorm: bad dims C=%d", C);
  FF_CHECK_ARG(rows < (1LL << 31), "ff_layernorm: too many rows");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const int nv = (C + 63) / 64;
#define LN_CASE(N) hipLaunchKernelGGL(layernorm_kernel<N>, grid, block, 0, st, in, ldi, out, ldo, (int)rows, C, gamma, beta, eps)
  if (nv <= 1) LN_CASE(1);
  else if (nv <= 2) LN_CASE(2);
  else if (nv <= 3) LN_CASE(3);
  else if (nv <= 4) LN_CASE(4);
  else if (nv <= 6) LN_CASE(6);
  else if (nv <= 8) LN_CASE(8);
  else LN_CASE(16);
#undef LN_CASE
  FF_LAUNCH_CHECK("ff_layernorm");
  return FF_OK;
}

// ---------------------------------------------------------------------------------------------
// pool: stage 1 = per-chunk column sums, stage 2 = sum of chunks / P.  Deterministic (no atomics).
__global__ __launch_bounds__(256) void pool_partial_kernel(const float* __restrict__ in, int ld, long long P, int C,
                                                           int pix_per_chunk, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
  const int chunk = blockIdx.x, b = blockIdx.y, nch = gridDim.x;
  const long long p0 = (long long)chunk * pix_per_chunk;
  long long p1 = p0 + pix_per_chunk;
  if (p1 > P) p1 = P;
  const float* base = in + (long long)b * P * ld;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    float s = 0.f;
    if (c < C)
      for (long long pp = p0 + wy; pp < p1; pp += 4) s += base[pp * ld + c];
    red[wy][lane] = s;
    __syncthreads();
    if (wy == 0 && c < C) part[((long long)b * nch + chunk) * C + c] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    __syncthreads();
  }
}

__global__ void pool_final_kernel(const float* __restrict__ part, int nch, int C, float invP, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  float s = 0.f;
  for (int i = 0; i < nch; ++i) s += part[((long long)b * nch + i) * C + c];
  out[(long long)b * C + c] = s * invP;
}

extern "C" int ff_pool_mean(const float* in, int ld, int B, long long P, int C, float* out, float* work,
                            long long work_floats, void* stream) {
  FF_CHECK_ARG(in && out && work, "ff_pool_mean: null pointer");
  FF_CHECK_ARG(B > 0 && P > 0 && C > 0 && ld >= C, "ff_pool_mean: bad dims");
  int nch = (int)((P + 255) / 256);
  if (nch > 512) nch = 512;
  const int ppc = (int)((P + nch - 1) / nch);
  nch = (int)((P + ppc - 1) / ppc);
  FF_CHECK_ARG(work_floats >= (long long)B * nch * C, "ff_pool_mean: workspace too small (need %lld floats)", (long long)B * nch * C);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pool_partial_kernel, dim3(nch, B), dim3(256), 0, st, in, ld, P, C, ppc, work);
  hipLaunchKernelGGL(pool_final_kernel, dim3((C + 255) / 256, B), dim3(256), 0, st, work, nch, C, 1.0f / (float)P, out);
  FF_LAUNCH_CHECK("ff_pool_mean");
  return FF_OK;
}

// workspace floats ff_pool_mean needs for (B, P, C)
extern "C" long long ff_pool_mean_workspace(int B, long long P, int C) {
  long long nch = (P + 255) / 256;
  if (nch > 512) nch = 512;
  return (long long)B * (nch + 1) * C;
}

// ---------------------------------------------------------------------------------------------
// out[b] = act2( W2 . act1( W1 . in[b] + b1 ) + b2 ) * post   (W2 == null: single layer, act1 is the output act)
__global__ __launch_bounds__(256) void vec_mlp_kernel(const float* __restrict__ in, int Cin, const float* __restrict__ W1,
                                                      const float* __restrict__ b1, int Ch, int act1,
                                                      const float* __restrict__ W2, const float* __restrict__ b2,
                                                      int Cout, int act2, float post, float* __restrict__ out) {
  extern __shared__ float sm[];
  float* xin = sm;            // [Cin]
  float* hid = sm + Cin;      // [Ch]
  const int b = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < Cin; i += 256) xin[i] = in[(long long)b * Cin + i];
  __syncthreads();
  if (W2) {
    for (int j = wid; j < Ch; j += 4) {
      float s = 0.f;
      for (int i = lane; i < Cin; i += 64) s += W1[(long long)j * Cin + i] * xin[i];
      s = wave_sum(s);
      if (lane == 0) hid[j] = ff_act(s + (b1 ? b1[j] : 0.f), act1);
    }
    __syncthreads();
    for (int o = blockIdx.x * 4 + wid; o < Cout; o += gridDim.x * 4) {
      float s = 0.f;
      for (int i = lane; i < Ch; i += 64) s += W2[(long long)o * Ch + i] * hid[i];
      s = wave_sum(s);
      if (lane == 0) out[(long long)b * Cout + o] = ff_act(s + (b2 ? b2[o] : 0.f), act2) * post;
    }
  } else {
    for (int o = blockIdx.x * 4 + wid; o < Ch; o += gridDim.x * 4) {
      float s = 0.f;
      for (int i = lane; i < Cin; i += 64) s += W1[(long long)o * Cin + i] * xin[i];
      s = wave_sum(s);
      if (lane == 0) out[(long long)b * Ch + o] = ff_act(s + (b1 ? b1[o] : 0.f), act1) * post;
    }
  }
}

extern "C" int ff_vec_mlp(const float* in, int B, int Cin, const float* W1, const float* b1, int Ch, int act1,
                          const float* W2, const float* b2, int Cout, int act2, float post, float* out, void* stream) {
  FF_CHECK_ARG(in && W1 && out, "ff_vec_mlp: null pointer");
  FF_CHECK_ARG(B > 0 && Cin > 0 && Ch > 0 && (Cin + Ch) * 4 <= 48 * 1024, "ff_vec_mlp: bad dims");
  FF_CHECK_ARG(!W2 || Cout > 0, "ff_vec_mlp: bad Cout");
  const int nout = W2 ? Cout : Ch;
  int gx = (nout + 3) / 4;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(vec_mlp_kernel, dim3(gx, B), dim3(256), (size_t)(Cin + Ch) * 4, (hipStream_t)stream, in, Cin, W1, b1, Ch,
                     act1, W2, b2, Cout, act2, post, out);
  FF_LAUNCH_CHECK("ff_vec_mlp");
  return FF_OK;
}

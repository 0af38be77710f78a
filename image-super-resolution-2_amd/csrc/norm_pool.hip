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
// Stage 1 (C % 4 == 0): a thread owns one float4 column and walks pixels with 4 independent loads in flight.
__global__ __launch_bounds__(256) void pool_partial_v4_kernel(const float* __restrict__ in, int ld, long long P, int C,
                                                              int pix_per_chunk, float* __restrict__ part) {
  extern __shared__ float4 red4[];                           // [rpi][cv]
  const int cv = C >> 2, rpi = 256 / cv;                     // float4 columns, pixel rows per iteration
  const int col = threadIdx.x % cv, row = threadIdx.x / cv;
  const int chunk = blockIdx.x, b = blockIdx.y, nch = gridDim.x;
  const long long p0 = (long long)chunk * pix_per_chunk;
  long long p1 = p0 + pix_per_chunk;
  if (p1 > P) p1 = P;
  const float* base = in + (long long)b * P * ld + 4 * col;
  float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (row < rpi) {
    long long pp = p0 + row;
    for (; pp + 3LL * rpi < p1; pp += 4LL * rpi) {
      const float4 v0 = *reinterpret_cast<const float4*>(base + pp * ld);
      const float4 v1 = *reinterpret_cast<const float4*>(base + (pp + rpi) * ld);
      const float4 v2 = *reinterpret_cast<const float4*>(base + (pp + 2LL * rpi) * ld);
      const float4 v3 = *reinterpret_cast<const float4*>(base + (pp + 3LL * rpi) * ld);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; pp < p1; pp += rpi) {
      const float4 v0 = *reinterpret_cast<const float4*>(base + pp * ld);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
    a0.x += a1.x + a2.x + a3.x; a0.y += a1.y + a2.y + a3.y; a0.z += a1.z + a2.z + a3.z; a0.w += a1.w + a2.w + a3.w;
    red4[row * cv + col] = a0;
  }
  __syncthreads();
  if (row == 0) {
    float4 s = red4[col];
    for (int r = 1; r < rpi; ++r) { const float4 t = red4[r * cv + col]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    *reinterpret_cast<float4*>(part + ((long long)b * nch + chunk) * C + 4 * col) = s;
  }
}

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

// stage 2: 64 channels x 4 partial-lanes per workgroup, 4 independent accumulators per thread
__global__ __launch_bounds__(256) void pool_final_kernel(const float* __restrict__ part, int nch, int C, int ldp, float invP,
                                                         float* __restrict__ out) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, b = blockIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    const float* pp = part + (long long)b * nch * ldp + c;
    int i = r;
    for (; i + 12 < nch; i += 16) {
      s0 += pp[(long long)i * ldp];
      s1 += pp[(long long)(i + 4) * ldp];
      s2 += pp[(long long)(i + 8) * ldp];
      s3 += pp[(long long)(i + 12) * ldp];
    }
    for (; i < nch; i += 4) s0 += pp[(long long)i * ldp];
  }
  red[r][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (r == 0 && c < C) out[(long long)b * C + c] = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * invP;
}

extern "C" int ff_pool_mean(const float* in, int ld, int B, long long P, int C, float* out, float* work,
                            long long work_floats, void* stream) {
  FF_CHECK_ARG(in && out && work, "ff_pool_mean: null pointer");
  FF_CHECK_ARG(B > 0 && P > 0 && C > 0 && ld >= C, "ff_pool_mean: bad dims");
  int nch = (int)((P + 255) / 256);
  if (nch > 1024) nch = 1024;
  const int ppc = (int)((P + nch - 1) / nch);
  nch = (int)((P + ppc - 1) / ppc);
  FF_CHECK_ARG(work_floats >= (long long)B * nch * C, "ff_pool_mean: workspace too small (need %lld floats)", (long long)B * nch * C);
  hipStream_t st = (hipStream_t)stream;
  const bool v4 = (C % 4 == 0) && (ld % 4 == 0) && (C / 4 <= 256) && (((uintptr_t)in & 15) == 0) && (((uintptr_t)work & 15) == 0);
  if (v4) {
    const int cv = C / 4, rpi = 256 / cv;
    hipLaunchKernelGGL(pool_partial_v4_kernel, dim3(nch, B), dim3(256), (size_t)rpi * cv * 16, st, in, ld, P, C, ppc, work);
  } else {
    hipLaunchKernelGGL(pool_partial_kernel, dim3(nch, B), dim3(256), 0, st, in, ld, P, C, ppc, work);
  }
  hipLaunchKernelGGL(pool_final_kernel, dim3((C + 63) / 64, B), dim3(256), 0, st, work, nch, C, C, 1.0f / (float)P, out);
  FF_LAUNCH_CHECK("ff_pool_mean");
  return FF_OK;
}

// first stage alone (one image): ff_pool_partial_rows(P) rows of C partial sums, for ff_pool_vec_mlp / ff_pool_finish
static int pool_chunks(long long P, int* ppc_out) {
  int nch = (int)((P + 255) / 256);
  if (nch > 1024) nch = 1024;
  const int ppc = (int)((P + nch - 1) / nch);
  if (ppc_out) *ppc_out = ppc;
  return (int)((P + ppc - 1) / ppc);
}
extern "C" int ff_pool_partial_rows(long long P) { return P > 0 ? pool_chunks(P, nullptr) : 0; }
extern "C" int ff_pool_partials(const float* in, int ld, long long P, int C, float* part, long long part_floats, void* stream) {
  FF_CHECK_ARG(in && part && P > 0 && C > 0 && ld >= C, "ff_pool_partials: bad args");
  int ppc;
  const int nch = pool_chunks(P, &ppc);
  FF_CHECK_ARG(part_floats >= (long long)nch * C, "ff_pool_partials: need %lld floats", (long long)nch * C);
  const bool v4 = (C % 4 == 0) && (ld % 4 == 0) && (C / 4 <= 256) && (((uintptr_t)in & 15) == 0) && (((uintptr_t)part & 15) == 0);
  if (v4) {
    const int cv = C / 4, rpi = 256 / cv;
    hipLaunchKernelGGL(pool_partial_v4_kernel, dim3(nch, 1), dim3(256), (size_t)rpi * cv * 16, (hipStream_t)stream, in, ld, P, C, ppc, part);
  } else {
    hipLaunchKernelGGL(pool_partial_kernel, dim3(nch, 1), dim3(256), 0, (hipStream_t)stream, in, ld, P, C, ppc, part);
  }
  FF_LAUNCH_CHECK("ff_pool_partials");
  return FF_OK;
}

// second stage alone: out[c] = inv_count * sum_r part[r][c] over `rows` rows of pitch ld (per-workgroup partials written by a
// producer kernel's epilogue, e.g. ff_conv3x3_halo pool_partials)
extern "C" int ff_pool_finish(const float* part, int rows, int ld, int C, float inv_count, float* out, void* stream) {
  FF_CHECK_ARG(part && out && rows > 0 && C > 0 && ld >= C, "ff_pool_finish: bad args");
  hipLaunchKernelGGL(pool_final_kernel, dim3((C + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream, part, rows, C, ld, inv_count, out);
  FF_LAUNCH_CHECK("ff_pool_finish");
  return FF_OK;
}

// workspace floats ff_pool_mean needs for (B, P, C)
extern "C" long long ff_pool_mean_workspace(int B, long long P, int C) {
  long long nch = (P + 255) / 256;
  if (nch > 1024) nch = 1024;
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
    const bool v4 = (Cin & 3) == 0 && (((uintptr_t)W1) & 15) == 0;       // NAFNet's SCA at 512 / 1024 channels: 16-byte weight loads
    for (int o = blockIdx.x * 4 + wid; o < Ch; o += gridDim.x * 4) {
      float s = 0.f;
      if (v4) {
        for (int i = 4 * lane; i < Cin; i += 256) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(W1 + (long long)o * Cin + i);
          s += (wv[0] * xin[i] + wv[1] * xin[i + 1]) + (wv[2] * xin[i + 2] + wv[3] * xin[i + 3]);
        }
      } else {
        for (int i = lane; i < Cin; i += 64) s += W1[(long long)o * Cin + i] * xin[i];
      }
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
  if (gx > 256) gx = 256;                          // one wave per output row up to 1024 rows: the 4 MB SCA matrix is read by every CU at once
  hipLaunchKernelGGL(vec_mlp_kernel, dim3(gx, B), dim3(256), (size_t)(Cin + Ch) * 4, (hipStream_t)stream, in, Cin, W1, b1, Ch,
                     act1, W2, b2, Cout, act2, post, out);
  FF_LAUNCH_CHECK("ff_vec_mlp");
  return FF_OK;
}

// ---------------------------------------------------------------------------------------------
// Pool finish + two-layer vector MLP in ONE single-workgroup launch (channel attention behind a conv's pool partials,
// hat_arch.py:50-54):  v[c] = inv_count * sum_r part[r][c];  out = act2(W2 . act1(W1 . v + b1) + b2) * post.
// Replaces ff_pool_finish + ff_vec_mlp (two dependent ~7 us launches on a block's critical path).  1024 threads: 16 row groups
// x 64 channel lanes for the reduction (4 independent accumulators per thread), then a wave per hidden unit, a thread per output.
__global__ __launch_bounds__(1024) void pool_vec_mlp_kernel(const float* __restrict__ part, int rows, int ld, float inv, int Cin,
                                                            const float* __restrict__ W1, const float* __restrict__ b1, int Ch, int act1,
                                                            const float* __restrict__ W2, const float* __restrict__ b2, int Cout, int act2,
                                                            float post, float* __restrict__ out, float* __restrict__ pooled) {
  extern __shared__ float sm[];
  float* red = sm;                       // [16][CP]  (CP = Cin rounded up to 64)
  const int CP = (Cin + 63) & ~63;
  float* xin = sm + 16 * CP;             // [Cin]
  float* hid = xin + Cin;                // [Ch]
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  // every 64-channel chunk of the row group's partial sums is accumulated in the same pass (independent loads in flight for all of
  // them), one barrier for the whole reduction
  // a single workgroup is latency bound: the loads of eight rows x four 64-channel chunks are issued before any is consumed
  // (one L2 round trip per 32 loads instead of one per row)
  for (int c0 = 0; c0 < CP; c0 += 256) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = rg; r0 < rows; r0 += 128) {
      float v[8][4];
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = r0 + 16 * j, c = c0 + 64 * k + lane;
          v[j][k] = (r < rows && c < Cin) ? part[(long long)r * ld + c] : 0.f;
        }
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += v[j][k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (c0 + 64 * k < CP) red[rg * CP + c0 + 64 * k + lane] = s[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Cin; c += 1024) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k * CP + c];
    t *= inv;
    xin[c] = t;
    if (pooled) pooled[c] = t;
  }
  __syncthreads();
  for (int j = rg; j < Ch; j += 16) {
    float s = 0.f;
    for (int i = lane; i < Cin; i += 64) s += W1[(long long)j * Cin + i] * xin[i];
    s = wave_sum(s);
    if (lane == 0) hid[j] = ff_act(s + (b1 ? b1[j] : 0.f), act1);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < Cout; o += 1024) {
    float s = b2 ? b2[o] : 0.f;
    for (int i = 0; i < Ch; ++i) s += W2[(long long)o * Ch + i] * hid[i];
    out[o] = ff_act(s, act2) * post;
  }
}

extern "C" int ff_pool_vec_mlp(const float* part, int rows, int ld, float inv_count, int Cin, const float* W1, const float* b1, int Ch,
                               int act1, const float* W2, const float* b2, int Cout, int act2, float post, float* out,
                               float* pooled_out, void* stream) {
  FF_CHECK_ARG(part && W1 && W2 && out, "ff_pool_vec_mlp: null pointer");
  FF_CHECK_ARG(rows > 0 && Cin > 0 && ld >= Cin && Ch > 0 && Ch <= 64 && Cout > 0 && Cin <= 512, "ff_pool_vec_mlp: bad dims (Cin <= 512, hidden width <= 64)");
  hipLaunchKernelGGL(pool_vec_mlp_kernel, dim3(1), dim3(1024), (size_t)(16 * ((Cin + 63) & ~63) + Cin + Ch) * 4, (hipStream_t)stream, part, rows, ld,
                     inv_count, Cin, W1, b1, Ch, act1, W2, b2, Cout, act2, post, out, pooled_out);
  FF_LAUNCH_CHECK("ff_pool_vec_mlp");
  return FF_OK;
}

// ---------------------------------------------------------------------------------------------
// Per-pixel two-layer MLP with a tiny hidden width and ONE output:  out[p] = act2( w2 . act1( W1 x[p] + b1 ) + b2 )
// (DAT AdaptiveInteraction spatial gate, dat_arch.py:585-590: conv1x1 180->11 (+BN folded) -> GELU -> conv1x1 11->1 -> sigmoid).
// As two GEMM launches this reads the 47 MB token tensor for 2 GFLOP and then runs a latency-bound 11->1 layer; here 16
// lanes own one pixel (three float4 of its channels each, 256-byte coalesced loads), a lane keeps 4 pixels in registers so
// every W1 row read from LDS is used four times, partial dot products are reduced across the 16 lanes, fp32 throughout.
#define PM_MAXH 16
// sum over the 16 lanes of a DPP row, result in every lane: two quad permutes, then the half-row and row mirrors (all lanes of
// a quad / half already hold the same partial sum when they are mirrored).  DPP modifiers run at VALU rate; __shfl_xor would be
// an LDS ds_bpermute per step.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}
__global__ __launch_bounds__(256) void pixel_mlp_kernel(const float* __restrict__ in, int ldi, long long P, int C, int Hd,
                                                        const float* __restrict__ W1, const float* __restrict__ b1, int act1,
                                                        const float* __restrict__ w2, float b2, int act2, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float W1s[PM_MAXH * 192];
  for (int i = threadIdx.x; i < PM_MAXH * 192; i += 256) {
    const int j = i / 192, c = i - j * 192;
    W1s[i] = (j < Hd && c < C) ? W1[j * C + c] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 4, j16 = lane & 15;
  const long long nwave = (long long)gridDim.x * 4;
  for (long long base = ((long long)blockIdx.x * 4 + wid) * 16; base < P; base += nwave * 16) {
    f32x4 x[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long pp = base + g + 4 * i;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = 4 * (j16 + 16 * k);
        const bool ok = pp < P && c < C;
        const f32x4 u = *reinterpret_cast<const f32x4*>(in + (ok ? pp * ldi + c : 0));
        x[i][k] = ok ? u : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    float o[4] = {b2, b2, b2, b2};
#pragma unroll 1
    for (int j = 0; j < Hd; ++j) {
      f32x4 wv[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) wv[k] = *reinterpret_cast<const f32x4*>(W1s + j * 192 + 4 * (j16 + 16 * k));
      const float bj = b1 ? b1[j] : 0.f, wj = w2[j];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 a = x[i][0] * wv[0];
        a += x[i][1] * wv[1];
        a += x[i][2] * wv[2];
        const float s = row16_sum((a[0] + a[1]) + (a[2] + a[3]));
        o[i] = __builtin_fmaf(wj, ff_act_fast(s + bj, act1), o[i]);   // A&S erf GELU (|err| <= 1.5e-7), as in the MFMA epilogues
      }
    }
    if (j16 == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long pp = base + g + 4 * i;
        if (pp < P) out[pp] = ff_act(o[i], act2);
      }
    }
  }
}

// C <= 64: one LANE per pixel (the 16-lanes-per-pixel form above leaves 8 of 16 lanes idle at C = 32 and ran the hierarchical
// fusion's SpatialGate at 0.9 TB/s): a lane reads its pixel's whole row (consecutive lanes = consecutive rows: coalesced when
// the rows are dense), the weights are wave-uniform scalar loads.
template <int C4>
__global__ __launch_bounds__(256) void pixel_mlp_lane_kernel(const float* __restrict__ in, int ldi, long long P, int Hd,
                                                             const float* __restrict__ W1, const float* __restrict__ b1, int act1,
                                                             const float* __restrict__ w2, float b2, int act2, float* __restrict__ out) {
  for (long long pp = (long long)blockIdx.x * 256 + threadIdx.x; pp < P; pp += (long long)gridDim.x * 256) {
    f32x4 x[C4];
#pragma unroll
    for (int k = 0; k < C4; ++k) x[k] = *reinterpret_cast<const f32x4*>(in + pp * ldi + 4 * k);
    float o = b2;
    for (int j = 0; j < Hd; ++j) {
      const float* wr = W1 + j * (4 * C4);
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int k = 0; k < C4; ++k) {
        s0 = __builtin_fmaf(x[k][0], wr[4 * k], s0); s1 = __builtin_fmaf(x[k][1], wr[4 * k + 1], s1);
        s0 = __builtin_fmaf(x[k][2], wr[4 * k + 2], s0); s1 = __builtin_fmaf(x[k][3], wr[4 * k + 3], s1);
      }
      o = __builtin_fmaf(w2[j], ff_act_fast(s0 + s1 + (b1 ? b1[j] : 0.f), act1), o);
    }
    out[pp] = ff_act(o, act2);
  }
}

extern "C" int ff_pixel_mlp(const float* in, int ldi, long long P, int C, int hidden, const float* W1, const float* b1, int act1,
                            const float* w2, float b2, int act2, float* out, void* stream) {
  FF_CHECK_ARG(in && W1 && w2 && out, "ff_pixel_mlp: null pointer");
  FF_CHECK_ARG(P > 0 && C > 0 && C <= 192 && C % 4 == 0 && ldi >= C && ldi % 4 == 0 && (((uintptr_t)in) & 15) == 0, "ff_pixel_mlp: needs C <= 192, C %% 4 == 0, 16-byte aligned rows");
  FF_CHECK_ARG(hidden > 0 && hidden <= PM_MAXH, "ff_pixel_mlp: hidden width must be 1..16");
  if (C == 32 || C == 64) {
    long long nbl = (P + 255) / 256;
    if (nbl > 16384) nbl = 16384;
    if (C == 32) hipLaunchKernelGGL(pixel_mlp_lane_kernel<8>, dim3((unsigned)nbl), dim3(256), 0, (hipStream_t)stream, in, ldi, P, hidden, W1, b1, act1, w2, b2, act2, out);
    else hipLaunchKernelGGL(pixel_mlp_lane_kernel<16>, dim3((unsigned)nbl), dim3(256), 0, (hipStream_t)stream, in, ldi, P, hidden, W1, b1, act1, w2, b2, act2, out);
    FF_LAUNCH_CHECK("ff_pixel_mlp");
    return FF_OK;
  }
  long long nb = (P + 63) / 64;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(pixel_mlp_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, ldi, P, C, hidden, W1, b1, act1, w2, b2, act2, out);
  FF_LAUNCH_CHECK("ff_pixel_mlp");
  return FF_OK;
}

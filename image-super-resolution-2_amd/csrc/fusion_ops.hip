// Small fused kernels of the fusion stack and the DAT channel attention statistics.
//   ff_chan_gram / ff_chan_attn_weights  dat_arch.py:627-647  (L2-normalised q,k over ALL tokens, 30x30 gram
//                                        per head, softmax) -> block-diagonal 180x180 matrix fed to ff_conv2d
//   ff_band_mha_core                     large_kernel_attention.py:222-224 (9-token, 4-head attention per pixel) and :393 (3 experts, 8 heads)
//   ff_band_weight                       multi_domain_frequency.py:498-503
//   ff_freq_guidance                     enhanced_fusion.py:533-542
//   ff_dynamic_gates                     fusion_network.py:226-234
//   ff_fuse_blend                        enhanced_fusion.py:550-556, 621-645 (one pass over the HR image)
#include "ff_common.h"

// ---------------------------------------------------------------------------------------------
// Channel-attention statistics.  qkv [N][ld] with q at q_off, k at k_off (heads*d channels each).
// Each workgroup reduces TOK tokens: per head a d x d gram (thread = 3x5 sub-tile) + column sums of squares.
#define CA_TOK 32
__global__ __launch_bounds__(384) void chan_gram_kernel(const float* __restrict__ qkv, int ld, int q_off, int k_off,
                                                        long long N, int tok_per_blk, float* __restrict__ part) {
  __shared__ float sq[CA_TOK][180], sk[CA_TOK][180];
  const int tid = threadIdx.x;
  const int head = tid / 60, sub = tid % 60, ti = (sub / 6) * 3, tj = (sub % 6) * 5;   // 10 x 6 sub-tiles of 3 x 5
  float g[3][5];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 5; ++b) g[a][b] = 0.f;
  float nrm = 0.f;                                         // threads 0..179: sum q^2 ; 180..359: sum k^2
  const long long n0 = (long long)blockIdx.x * tok_per_blk;
  long long n1 = n0 + tok_per_blk;
  if (n1 > N) n1 = N;
  for (long long base = n0; base < n1; base += CA_TOK) {
    const int cnt = (int)((n1 - base) < CA_TOK ? (n1 - base) : CA_TOK);
    for (int i = tid; i < CA_TOK * 180; i += 384) {
      const int t = i / 180, c = i % 180;
      float qv = 0.f, kv = 0.f;
      if (t < cnt) { qv = qkv[(base + t) * ld + q_off + c]; kv = qkv[(base + t) * ld + k_off + c]; }
      sq[t][c] = qv; sk[t][c] = kv;
    }
    __syncthreads();
    if (tid < 360) {
#pragma unroll 4
      for (int t = 0; t < CA_TOK; ++t) {
        float qa[3], kb[5];
#pragma unroll
        for (int a = 0; a < 3; ++a) qa[a] = sq[t][head * 30 + ti + a];
#pragma unroll
        for (int b = 0; b < 5; ++b) kb[b] = sk[t][head * 30 + tj + b];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 5; ++b) g[a][b] += qa[a] * kb[b];
        const float v = tid < 180 ? sq[t][tid] : sk[t][tid - 180];
        nrm += v * v;
      }
    }
    __syncthreads();
  }
  if (tid < 360) {
    float* o = part + (long long)blockIdx.x * 5760;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 5; ++b) o[head * 900 + (ti + a) * 30 + tj + b] = g[a][b];
    o[5400 + tid] = nrm;
  }
}

// stage 2: 180 workgroups of 32 statistics x 8 block slices: a thread sums every 8th partial block of its statistic (four independent
// accumulators: 8 dependent load rounds at nblk = 256 instead of 64), the slices meet in LDS in a fixed order (deterministic)
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float* __restrict__ part, int nblk, float* __restrict__ G) {
  __shared__ float red[8][32];
  const int l = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + l;                       // 5760 = 180 x 32
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = sl;
  for (; b + 24 < nblk; b += 32) {
    s0 += part[(long long)b * 5760 + e];
    s1 += part[(long long)(b + 8) * 5760 + e];
    s2 += part[(long long)(b + 16) * 5760 + e];
    s3 += part[(long long)(b + 24) * 5760 + e];
  }
  for (; b < nblk; b += 8) s0 += part[(long long)b * 5760 + e];
  red[sl][l] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][l];
    G[e] = t;
  }
}

// stage 3: normalise, softmax rows, emit block-diagonal weight Wbd[co][ci] (one thread per (row, column))
__global__ __launch_bounds__(256) void chan_attn_weights_kernel(const float* __restrict__ G,
                                                                const float* __restrict__ temperature,
                                                                float* __restrict__ wbd) {
  const int row = blockIdx.x;                                // head*30 + i
  const int head = row / 30, i = row % 30;
  __shared__ float v[30];
  const float nq = fmaxf(sqrtf(G[5400 + row]), 1e-12f);
  if (threadIdx.x < 30) {
    const int j = threadIdx.x;
    const float nk = fmaxf(sqrtf(G[5400 + 180 + head * 30 + j]), 1e-12f);
    v[j] = G[head * 900 + i * 30 + j] / (nq * nk) * temperature[head];
  }
  __syncthreads();
  float mx = -INFINITY, s = 0.f;
  for (int j = 0; j < 30; ++j) mx = fmaxf(mx, v[j]);
  for (int j = 0; j < 30; ++j) s += expf(v[j] - mx);
  for (int c = threadIdx.x; c < 180; c += blockDim.x) {
    const int j = c - head * 30;
    wbd[row * 180 + c] = (j >= 0 && j < 30) ? expf(v[j] - mx) / s : 0.f;
  }
}

extern "C" int ff_chan_attn_weights(const float* qkv, int ld, int q_off, int k_off, long long N, const float* temperature,
                                    float* wbd, float* work, long long work_floats, void* stream) {
  FF_CHECK_ARG(qkv && temperature && wbd && work && N > 0, "ff_chan_attn_weights: bad args");
  int nblk = (int)((N + 255) / 256);
  if (nblk > 1024) nblk = 1024;
  const int tpb = (int)((N + nblk - 1) / nblk);
  nblk = (int)((N + tpb - 1) / tpb);
  FF_CHECK_ARG(work_floats >= (long long)(nblk + 1) * 5760, "ff_chan_attn_weights: workspace too small (need %lld floats)", (long long)(nblk + 1) * 5760);
  hipStream_t st = (hipStream_t)stream;
  float* G = work + (long long)nblk * 5760;
  hipLaunchKernelGGL(chan_gram_kernel, dim3(nblk), dim3(384), 0, st, qkv, ld, q_off, k_off, N, tpb, work);
  hipLaunchKernelGGL(chan_reduce_kernel, dim3(5760 / 32), dim3(256), 0, st, work, nblk, G);
  hipLaunchKernelGGL(chan_attn_weights_kernel, dim3(180), dim3(64), 0, st, G, temperature, wbd);
  FF_LAUNCH_CHECK("ff_chan_attn_weights");
  return FF_OK;
}

// reduce + normalise + softmax of per-workgroup partials already in `work` (written by ff_chan_qkv: nblk blocks of 5760 floats)
extern "C" int ff_chan_attn_finish(float* work, long long work_floats, int nblk, const float* temperature, float* wbd, void* stream) {
  FF_CHECK_ARG(work && temperature && wbd && nblk > 0, "ff_chan_attn_finish: bad args");
  FF_CHECK_ARG(work_floats >= (long long)(nblk + 1) * 5760, "ff_chan_attn_finish: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* G = work + (long long)nblk * 5760;
  hipLaunchKernelGGL(chan_reduce_kernel, dim3(5760 / 32), dim3(256), 0, st, work, nblk, G);
  hipLaunchKernelGGL(chan_attn_weights_kernel, dim3(180), dim3(64), 0, st, G, temperature, wbd);
  FF_LAUNCH_CHECK("ff_chan_attn_finish");
  return FF_OK;
}

extern "C" long long ff_chan_attn_workspace(long long N) {
  long long nblk = (N + 255) / 256;
  if (nblk > 1024) nblk = 1024;
  return (nblk + 2) * 5760;
}

// ---------------------------------------------------------------------------------------------
// 9-token / 4-head / d=16 attention per pixel.  qkv rows [(p*9 + i)][192] = q|k|v; out rows [(p*9+i)][64].
template <int NB>
__global__ __launch_bounds__(256) void band_mha_core_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            long long P, int heads) {
  constexpr int nb = NB;
  const int d = 16, E = heads * d;
  const long long total = P * nb * heads;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int h = (int)(idx % heads);
    long long t = idx / heads;
    const int i = (int)(t % nb);
    const long long p = t / nb;
    const float* base = qkv + p * nb * 3 * E;
    f32x4 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const f32x4*>(base + (long long)i * 3 * E + h * d + 4 * u) * 0.25f;
    float s[NB];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < nb; ++j) {
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
    for (int j = 0; j < nb; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
    f32x4 o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < nb; ++j) {
      const float w = s[j] / den;
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] += *reinterpret_cast<const f32x4*>(base + (long long)j * 3 * E + 2 * E + h * d + 4 * u) * w;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(out + (p * nb + i) * E + h * d + 4 * u) = o[u];
  }
}

extern "C" int ff_band_mha_core(const float* qkv, float* out, long long P, int nbands, int heads, void* stream) {
  FF_CHECK_ARG(qkv && out && P > 0 && heads > 0, "ff_band_mha_core: bad args");
  FF_CHECK_ARG(nbands == 9 || nbands == 3, "ff_band_mha_core: built for 9 tokens (frequency bands) or 3 (experts), got %d", nbands);
  long long nb = (P * nbands * heads + 255) / 256;
  if (nb > 16384) nb = 16384;
  if (nbands == 9) hipLaunchKernelGGL(band_mha_core_kernel<9>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, qkv, out, P, heads);
  else hipLaunchKernelGGL(band_mha_core_kernel<3>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, qkv, out, P, heads);
  FF_LAUNCH_CHECK("ff_band_mha_core");
  return FF_OK;
}

// ---------------------------------------------------------------------------------------------
// out[p][3i+c] = x[p][3i+c] * att[p][i] * imp[i]
__global__ __launch_bounds__(256) void band_weight_kernel(const float* __restrict__ x, const float* __restrict__ att,
                                                          const float* __restrict__ imp, float* __restrict__ out,
                                                          long long P, int nb) {
  const int C = nb * 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P * C; i += (long long)gridDim.x * 256) {
    const long long p = i / C;
    const int b = (int)(i - p * C) / 3;
    out[i] = x[i] * att[p * nb + b] * imp[b];
  }
}

extern "C" int ff_band_weight(const float* x, const float* att, const float* imp, float* out, long long P, int nbands,
                              void* stream) {
  FF_CHECK_ARG(x && att && imp && out && P > 0 && nbands > 0, "ff_band_weight: bad args");
  long long nb = (P * nbands * 3 + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(band_weight_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, att, imp, out, P, nbands);
  FF_LAUNCH_CHECK("ff_band_weight");
  return FF_OK;
}

// bands3 [P][9] = (low rgb, mid rgb, high rgb)  ->  guide [P][3] = (high, mid, low) / (sum + 1e-8)
__global__ __launch_bounds__(256) void freq_guidance_kernel(const float* __restrict__ b3, float* __restrict__ guide, long long P) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
    const float* s = b3 + p * 9;
    const float lo = (fabsf(s[0]) + fabsf(s[1]) + fabsf(s[2])) / 3.f;
    const float mi = (fabsf(s[3]) + fabsf(s[4]) + fabsf(s[5])) / 3.f;
    const float hi = (fabsf(s[6]) + fabsf(s[7]) + fabsf(s[8])) / 3.f;
    const float tot = lo + mi + hi + 1e-8f;
    guide[p * 3] = hi / tot; guide[p * 3 + 1] = mi / tot; guide[p * 3 + 2] = lo / tot;
  }
}

extern "C" int ff_freq_guidance(const float* bands3, float* guide, long long P, void* stream) {
  FF_CHECK_ARG(bands3 && guide && P > 0, "ff_freq_guidance: bad args");
  long long nb = (P + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(freq_guidance_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, bands3, guide, P);
  FF_LAUNCH_CHECK("ff_freq_guidance");
  return FF_OK;
}

// graw [P][3] (already sigmoid), dif [P] -> gates [P][3]
__global__ __launch_bounds__(256) void dynamic_gates_kernel(const float* __restrict__ graw, const float* __restrict__ dif,
                                                            float* __restrict__ gates, long long P) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
    const float th = 0.7f - 0.4f * dif[p];
    float g[3], mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      g[e] = 1.0f / (1.0f + expf(-10.0f * (graw[p * 3 + e] - th)));
      mx = fmaxf(mx, g[e]);
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) gates[p * 3 + e] = fmaxf(g[e], (g[e] >= mx * 0.99f) ? 0.9f : 0.f);
  }
}

extern "C" int ff_dynamic_gates(const float* graw, const float* dif, float* gates, long long P, void* stream) {
  FF_CHECK_ARG(graw && dif && gates && P > 0, "ff_dynamic_gates: bad args");
  long long nb = (P + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(dynamic_gates_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, graw, dif, gates, P);
  FF_LAUNCH_CHECK("ff_dynamic_gates");
  return FF_OK;
}

// HR blend: experts E [Hh][Wh][9] (hat,dat,naf rgb), hier [Hh][Wh][3]; LR maps guide/gates [Hl][Wl][3], dif [Hl][Wl]
__device__ __forceinline__ void bl_setup(int o, int n_in, int n_out, int& i0, int& i1, float& l) {
  if (n_in == n_out) { i0 = i1 = o; l = 0.f; return; }
  const float sc = (float)n_in / (float)n_out;
  const float s = fmaxf(sc * ((float)o + 0.5f) - 0.5f, 0.f);
  i0 = min((int)floorf(s), n_in - 1); i1 = min(i0 + 1, n_in - 1);
  l = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
}

__global__ __launch_bounds__(256) void fuse_blend_kernel(const float* __restrict__ E, const float* __restrict__ hier,
                                                         const float* __restrict__ guide, const float* __restrict__ gates,
                                                         const float* __restrict__ dif, float* __restrict__ out, int Hh, int Wh,
                                                         int Hl, int Wl) {
  const long long P = (long long)Hh * Wh;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
    const int x = (int)(p % Wh), y = (int)(p / Wh);
    int y0, y1, x0, x1; float ly, lx;
    bl_setup(y, Hl, Hh, y0, y1, ly);
    bl_setup(x, Wl, Wh, x0, x1, lx);
    const long long a = (long long)y0 * Wl + x0, b = (long long)y0 * Wl + x1, c = (long long)y1 * Wl + x0, d = (long long)y1 * Wl + x1;
#define BL(T, k, s) ((1.f - ly) * ((1.f - lx) * T[a * s + k] + lx * T[b * s + k]) + ly * ((1.f - lx) * T[c * s + k] + lx * T[d * s + k]))
    float gd[3], gt[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) { gd[e] = BL(guide, e, 3); gt[e] = BL(gates, e, 3); }
    const float df = BL(dif, 0, 1);
#undef BL
    const float gsum = gt[0] + gt[1] + gt[2] + 1e-8f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float eh = E[p * 9 + ch], ed = E[p * 9 + 3 + ch], en = E[p * 9 + 6 + ch];
      const float fw = eh * gd[0] + ed * gd[1] + en * gd[2];
      const float f0 = hier[p * 3 + ch] * 0.7f + fw * 0.3f;
      const float dyn = (eh * gt[0] + ed * gt[1] + en * gt[2]) / gsum;
      out[p * 3 + ch] = f0 * (1.f - 0.3f * df) + dyn * (0.3f * df);
    }
  }
}

extern "C" int ff_fuse_blend(const float* experts9, const float* hier3, const float* guide3, const float* gates3,
                             const float* dif1, float* out3, int Hh, int Wh, int Hl, int Wl, void* stream) {
  FF_CHECK_ARG(experts9 && hier3 && guide3 && gates3 && dif1 && out3 && Hh > 0 && Wh > 0 && Hl > 0 && Wl > 0, "ff_fuse_blend: bad args");
  long long nb = ((long long)Hh * Wh + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(fuse_blend_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, experts9, hier3, guide3, gates3,
                     dif1, out3, Hh, Wh, Hl, Wl);
  FF_LAUNCH_CHECK("ff_fuse_blend");
  return FF_OK;
}

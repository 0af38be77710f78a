// DAT channel-attention front end in one launch (dat_arch.py:617-647, AdaptiveChannelAttention.forward up to the softmax):
//     qkv = Linear(LayerNorm(x));  per head h:  G_h = q_h^T k_h  (30 x 30, contraction over ALL tokens),  |q_h[:, i]|^2,  |k_h[:, j]|^2
// The two-stage form wrote q, k, v (141 MB) with ff_token_linear and read q, k back (94 MB) in a VALU gram kernel (80 us).
// Here q and k never reach memory: a wave keeps its 32 tokens' normalised rows as split-bf16 fragments (as ff_token_linear), computes
// the q_h and k_h tiles with the MFMA operands SWAPPED so the channel sits on the lane and the tokens in the accumulator registers
// -- registers 8s..8s+7 are then the operands of G_h += q_h^T k_h over the wave's tokens (guide section 3, 'An accumulator tile as
// the next MFMA's operand') -- reduces the eight waves' 32x32 partial grams through LDS in a fixed order (deterministic), and
// writes only v (which the depth-wise conv branch and the attention product need) plus one 5760-float partial per workgroup in
// the layout ff_chan_attn_finish reduces.
#include "ff_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ChanQkvParams {
  const float* x; const float* gamma; const float* beta;
  const __bf16* w;        // [18 tiles][2 planes][32][192]: q_0, k_0, ..., q_5, k_5 (head dim padded to 32 rows), then v rows 0..191 in 6 tiles
  const float* bias;      // [18*32] zero padded
  float* v; float* part;
  long long M;
  int ldx, ldv, K, C;
  float eps;
};

#define CQ_KS 12
#define CQ_SLOTS 25
#define CQ_ROWB (CQ_SLOTS * 16)
#define CQ_PL (32 * CQ_SLOTS)
#define CQ_PLB (CQ_PL * 16)
#define CQ_BUFB (2 * CQ_PLB)
#define CQ_TILE_ELEMS (32 * 192)
#define CQ_RED 1088                       // floats per wave in the reduction buffer: 32x32 gram + 32 q norms + 32 k norms
#define CQ_TR 36
#define CQ_OFF_RED (2 * CQ_BUFB)
#define CQ_OFF_TR (CQ_OFF_RED + 8 * CQ_RED * 4)
#define CQ_OFF_BS (CQ_OFF_TR + 8 * 32 * CQ_TR * 4)
#define CQ_LDS (CQ_OFF_BS + 18 * 32 * 4)

template <int NTERMS>
__global__ __launch_bounds__(512) void chan_qkv_kernel(ChanQkvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem + CQ_OFF_RED);
  float* Bs = reinterpret_cast<float*>(smem + CQ_OFF_BS);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  float* tr = reinterpret_cast<float*>(smem + CQ_OFF_TR) + wid * (32 * CQ_TR);
  float* xs = reinterpret_cast<float*>(smem + CQ_OFF_RED) + wid * (32 * FF_XS_ROW);      // prologue gather patch over red + tr
  const long long tok0 = (long long)blockIdx.x * 256 + wid * 32;
  const long long tok = tok0 + l31;
  const bool tvalid = tok < p.M;
  const unsigned vmask = (unsigned)(__ballot(tvalid) & 0xffffffffu);

  int off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int s = (wid + 8 * i) * 64 + lane;
    if (s >= 2 * CQ_PL) s = 2 * CQ_PL - 1;
    const int plane = s / CQ_PL, t = s - plane * CQ_PL, row = t / CQ_SLOTS;
    int q = t - row * CQ_SLOTS;
    if (q > 2 * CQ_KS - 1) q = 2 * CQ_KS - 1;
    off[i] = plane * CQ_TILE_ELEMS + row * 192 + q * 8;
  }
  auto dma = [&](int tile, int buf) {
    const __bf16* rec = p.w + (long long)tile * (2 * CQ_TILE_ELEMS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (wid + 8 * i < 25)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + off[i]),
                                         (__attribute__((address_space(3))) void*)(smem + buf * CQ_BUFB + (wid + 8 * i) * 1024), 16, 0, 0);
  };
  dma(0, 0);
  for (int i = tid; i < 18 * 32; i += 512) Bs[i] = p.bias ? p.bias[i] : 0.f;

  // ---- x rows -> LayerNorm -> split fragments -----------------------------------------------------------------------------
  bf16x8 xh[CQ_KS], xl[CQ_KS];
  {
    float v[CQ_KS][8];
    ff_wave_rows_to_frags<3>(p.x, p.ldx, tok0, p.M, p.K, xs, lane, v);
    float s = 0.f;
#pragma unroll
    for (int st = 0; st < CQ_KS; ++st)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[st][j];
    s += __shfl_xor(s, 32);
    const float mean = s / (float)p.K;
    float qv = 0.f;
#pragma unroll
    for (int st = 0; st < CQ_KS; ++st)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = (16 * st + 8 * hh + j < p.K) ? v[st][j] - mean : 0.f;
        qv += d * d;
      }
    qv += __shfl_xor(qv, 32);
    const float rstd = 1.0f / sqrtf(qv / (float)p.K + p.eps);
#pragma unroll
    for (int st = 0; st < CQ_KS; ++st) {
      const int k0 = 16 * st + 8 * hh;
      const int ka = k0 < p.K ? k0 : 0, kb = k0 + 4 < p.K ? k0 + 4 : 0;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + ka), b0 = *reinterpret_cast<const f32x4*>(p.beta + ka);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(p.gamma + kb), b1 = *reinterpret_cast<const f32x4*>(p.beta + kb);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gg = j < 4 ? g0[j & 3] : g1[j & 3], bb = j < 4 ? b0[j & 3] : b1[j & 3];
        const float f = (k0 + j < p.K) ? (v[st][j] - mean) * rstd * gg + bb : 0.f;
        const __bf16 h = (__bf16)f;
        xh[st][j] = h;
        xl[st][j] = (__bf16)(f - (float)h);
      }
    }
  }

  auto gemm_tile = [&](int buf, bool swap, f32x16& acc) {
    const unsigned char* ap = smem + buf * CQ_BUFB + l31 * CQ_ROWB + 16 * hh;
    bf16x8 fa[2], fl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      fa[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u);
      fl[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u + CQ_PLB);
    }
#pragma unroll
    for (int st = 0; st < CQ_KS; ++st) {
      const bf16x8 ah = fa[st & 1], al = fl[st & 1];
      if (st + 2 < CQ_KS) {
        fa[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2));
        fl[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2) + CQ_PLB);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!swap) {
        if (NTERMS == 3) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl[st], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[st], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[st], acc, 0, 0, 0);
      } else {
        if (NTERMS == 3) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[st], ah, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[st], al, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[st], ah, acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int it = 0;
  auto ring_step = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 1 < 18) dma(it + 1, (it + 1) & 1);
  };
  // swapped tile (lane = channel, registers = the wave's tokens): bias, zero the tokens beyond M, squared column norm,
  // split fragments whose registers 8s..8s+7 are k-step s of the gram product
  auto qk_tile = [&](bf16x8 (&fh)[2], bf16x8 (&fl)[2], float& nrm) {
    ring_step();
    f32x16 acc;
    const float bv = Bs[it * 32 + l31];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bv;
    gemm_tile(it & 1, true, acc);
    float n2 = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 8 * s + j;
        const float f = ((vmask >> ((r & 3) + 8 * (r >> 2) + 4 * hh)) & 1u) ? acc[r] : 0.f;
        n2 += f * f;
        const __bf16 h = (__bf16)f;
        fh[s][j] = h;
        fl[s][j] = (__bf16)(f - (float)h);
      }
    nrm = n2 + __shfl_xor(n2, 32);
    ++it;
  };

  float* po = p.part + (long long)blockIdx.x * 5760;
#pragma unroll 1
  for (int h = 0; h < 6; ++h) {
    bf16x8 qh[2], ql[2], kh[2], kl[2];
    float nq, nk;
    qk_tile(qh, ql, nq);
    qk_tile(kh, kl, nk);
    f32x16 g;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {                          // G[i][j] += sum over this wave's tokens of q[t][i] k[t][j]
      if (NTERMS == 3) {
        g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[s], kl[s], g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql[s], kh[s], g, 0, 0, 0);
      }
      g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[s], kh[s], g, 0, 0, 0);
    }
    float* rw = red + wid * CQ_RED;
#pragma unroll
    for (int r = 0; r < 16; ++r) rw[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + l31] = g[r];      // row i, column j = lane
    if (hh == 0) { rw[1024 + l31] = nq; rw[1056 + l31] = nk; }
    __syncthreads();
    // fixed-order sum over the eight waves: thread t owns elements t, t + 512 (and t + 1024 for t < 64)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      s0 += red[w * CQ_RED + tid];
      s1 += red[w * CQ_RED + 512 + tid];
      if (tid < 64) s2 += red[w * CQ_RED + 1024 + tid];
    }
    // this workgroup's partial in ff_chan_attn_finish's layout: [6][30][30] grams, then 180 q norms, 180 k norms
    {
      const int i0 = tid >> 5, j0 = tid & 31;
      if (i0 < 30 && j0 < 30) po[h * 900 + i0 * 30 + j0] = s0;
      if (i0 + 16 < 30 && j0 < 30) po[h * 900 + (i0 + 16) * 30 + j0] = s1;
      if (tid < 30) po[5400 + h * 30 + tid] = s2;
      else if (tid >= 32 && tid < 62) po[5400 + 180 + h * 30 + tid - 32] = s2;
    }
    __syncthreads();
  }
  // ---- v tiles: the standard orientation, stored as 128-byte row segments ------------------------------------------------------
#pragma unroll 1
  for (int t = 0; t < 6; ++t) {
    ring_step();
    f32x16 acc;
    const int nb = it * 32 + 4 * hh;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = Bs[nb + (r & 3) + 8 * (r >> 2)];
    gemm_tile(it & 1, false, acc);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 v4;
#pragma unroll
      for (int e = 0; e < 4; ++e) v4[e] = acc[4 * g4 + e];
      *reinterpret_cast<f32x4*>(tr + l31 * CQ_TR + 8 * g4 + 4 * hh) = v4;
    }
    const int tq = lane >> 3, c4 = t * 32 + 4 * (lane & 7);
    if (c4 < p.C) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 ov = *reinterpret_cast<const f32x4*>(tr + (tq + 8 * i) * CQ_TR + 4 * (lane & 7));
        if (tok0 + tq + 8 * i < p.M) *reinterpret_cast<f32x4*>(p.v + (tok0 + tq + 8 * i) * p.ldv + c4) = ov;
      }
    }
    ++it;
  }
}

extern "C" long long ff_chan_qkv_workspace(long long M) { return ((M + 255) / 256 + 2) * 5760; }

extern "C" int ff_chan_qkv(const float* x, int ldx, long long M, int K, const float* gamma, const float* beta, float eps,
                           const void* w_tiles, const float* bias_padded, float* v_out, int ldv, float* work, long long work_floats,
                           int nterms, void* stream) {
  FF_CHECK_ARG(nterms == 1 || nterms == 3, "ff_chan_qkv: nterms must be 1 or 3");
  FF_CHECK_ARG(x && gamma && beta && w_tiles && v_out && work && M > 0, "ff_chan_qkv: null pointer");
  FF_CHECK_ARG(K == 180 && ldx >= K && ldx % 4 == 0 && ldv >= K && ldv % 4 == 0, "ff_chan_qkv: built for DAT's 180 channels (6 heads of 30), 16-byte aligned rows");
  FF_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)v_out) | ((uintptr_t)w_tiles) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0, "ff_chan_qkv: 16-byte alignment");
  const long long nblk = (M + 255) / 256;
  FF_CHECK_ARG(nblk < (1LL << 31) && work_floats >= (nblk + 1) * 5760, "ff_chan_qkv: workspace too small (need %lld floats)", (nblk + 1) * 5760);
  ChanQkvParams p;
  p.x = x; p.gamma = gamma; p.beta = beta; p.w = (const __bf16*)w_tiles; p.bias = bias_padded; p.v = v_out; p.part = work;
  p.M = M; p.ldx = ldx; p.ldv = ldv; p.K = K; p.C = 180; p.eps = eps;
  static_assert(CQ_LDS <= 160 * 1024, "LDS image too large");
  static_assert(8 * 32 * FF_XS_ROW * 4 <= 8 * CQ_RED * 4 + 8 * 32 * CQ_TR * 4, "gather patch must fit in the reduction + transpose area");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chan_qkv_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, CQ_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chan_qkv_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, CQ_LDS);
    if (e != hipSuccess) { ff_set_error("ff_chan_qkv: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; }
    attr_set = true;
  }
  if (nterms == 3) hipLaunchKernelGGL(chan_qkv_kernel<3>, dim3((unsigned)nblk), dim3(512), CQ_LDS, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(chan_qkv_kernel<1>, dim3((unsigned)nblk), dim3(512), CQ_LDS, (hipStream_t)stream, p);
  FF_LAUNCH_CHECK("ff_chan_qkv");
  return FF_OK;
}

// NAFNet block fusions (nafnet_arch.py:110-131), all bandwidth-bound at HR resolution (1 048 576 pixels x 64 ch):
//   ff_dwconv3_gate_pool : conv2 (depth-wise 3x3 on 2c channels, :78-81) + SimpleGate (:51-52) + the per-channel sums of the
//                          SCA average pool (:85-88) in one pass: reads 2c, writes c (was: write 2c, read 2c, write c, read c)
//   ff_naf_ffn           : y + gamma * conv5( SimpleGate( conv4( LayerNorm2d(y) ) ) )  (:124-131) for c = 64 / 128 in one
//                          launch with the 2c-wide hidden activation kept on chip ("flash" structure of token_mlp.hip:
//                          LN(y) rows as register/LDS-resident MFMA B operands, W4 tile PAIRS (channels j and j+c) and W5
//                          tiles through an LDS-DMA ring, gate = product of the two H^T accumulator tiles, which then IS the
//                          B operand of the second GEMM).  HBM traffic: y in, out out (537 MB at level 0 instead of 2.9 GB).
#include "ff_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// --------------------------------------------------------------------------------------------- dwconv + gate + pool sums
// A thread owns 4 channels (of each gate half) of a 4-row output strip at one column: a 6 x 3 input window and the 9 tap
// weights per half live in registers, so an output costs 9 float4 input loads (18 with one pixel per thread) and the
// weights are fetched once per strip.  The two halves are evaluated one after the other to reuse the window registers.
__global__ __launch_bounds__(256) void dwconv3_gate_pool_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo,
                                                                int H, int W, int C, const float* __restrict__ w, const float* __restrict__ bias,
                                                                float* __restrict__ part) {
  extern __shared__ float4 red[];                      // [ppb][C/4]
  const int cv = C >> 2, ppb = 256 / cv;               // float4 channel groups, (column, strip) items per block iteration
  const int cg = threadIdx.x % cv, ps = threadIdx.x / cv;
  const int nstrip = (H + 3) >> 2;
  const long long items = (long long)nstrip * W;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc_sum = z4;
  if (ps < ppb) {
    const int c = 4 * cg;
    for (long long it = (long long)blockIdx.x * ppb + ps; it < items; it += (long long)gridDim.x * ppb) {
      const int x = (int)(it % W), y0 = (int)(it / W) * 4;
      f32x4 ga[4];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int ch = c + half * C;
        f32x4 wt[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[k] = *reinterpret_cast<const f32x4*>(w + k * 2 * C + ch);
        const f32x4 bs = *reinterpret_cast<const f32x4*>(bias + ch);
        f32x4 win[6][3];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const int iy = y0 - 1 + r;
          const bool oky = (unsigned)iy < (unsigned)H;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int ix = x - 1 + k;
            const bool ok = oky && (unsigned)ix < (unsigned)W;
            const f32x4 u = *reinterpret_cast<const f32x4*>(in + (ok ? ((long long)iy * W + ix) * ldi + ch : 0));
            win[r][k] = ok ? u : z4;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f32x4 a = bs;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a += win[r + ky][kx] * wt[ky * 3 + kx];
          ga[r] = half == 0 ? a : ga[r] * a;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (y0 + r < H) {
          *reinterpret_cast<f32x4*>(out + ((long long)(y0 + r) * W + x) * ldo + c) = ga[r];
          acc_sum += ga[r];
        }
    }
    red[ps * cv + cg] = (float4){acc_sum[0], acc_sum[1], acc_sum[2], acc_sum[3]};
  }
  __syncthreads();
  if (ps == 0) {
    float4 s = red[cg];
    for (int r = 1; r < ppb; ++r) { const float4 t = red[r * cv + cg]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    *reinterpret_cast<float4*>(part + (long long)blockIdx.x * C + 4 * cg) = s;
  }
}

// sum of the per-workgroup partials: 16 channels x 16 partial-lanes per workgroup, 4 independent accumulators per thread
__global__ __launch_bounds__(256) void pool_reduce_kernel(const float* __restrict__ part, int nch, int C, float invP, float* __restrict__ out) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int i = r;
    for (; i + 48 < nch; i += 64) {
      s0 += part[(long long)i * C + c];
      s1 += part[(long long)(i + 16) * C + c];
      s2 += part[(long long)(i + 32) * C + c];
      s3 += part[(long long)(i + 48) * C + c];
    }
    for (; i < nch; i += 16) s0 += part[(long long)i * C + c];
  }
  red[r][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (r == 0 && c < C) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][cl];
    out[c] = s * invP;
  }
}

extern "C" long long ff_dwconv3_gate_pool_workspace(int C) { return (long long)1024 * C; }

extern "C" int ff_dwconv3_gate_pool(const float* in, int ldi, float* out, int ldo, int H, int W, int C, const float* w_tapmajor,
                                    const float* bias, float* pooled, float* work, long long work_floats, void* stream) {
  FF_CHECK_ARG(in && out && w_tapmajor && bias && pooled && work, "ff_dwconv3_gate_pool: null pointer");
  FF_CHECK_ARG(H > 0 && W > 0 && C >= 4 && C % 4 == 0 && C <= 1024 && ldi >= 2 * C && ldi % 4 == 0 && ldo >= C && ldo % 4 == 0, "ff_dwconv3_gate_pool: bad dims");
  FF_CHECK_ARG((((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & 15) == 0 && (((uintptr_t)w_tapmajor) & 15) == 0 && (((uintptr_t)bias) & 15) == 0 && (((uintptr_t)work) & 15) == 0, "ff_dwconv3_gate_pool: 16-byte alignment required");
  const int cv = C / 4, ppb = 256 / cv;
  const long long P = (long long)H * W;
  long long nb = ((long long)((H + 3) / 4) * W + ppb - 1) / ppb;      // (column, 4-row strip) items
  if (nb > 1024) nb = 1024;      // 4 workgroups per CU: enough to stream at HBM rate, few enough partials to reduce
  FF_CHECK_ARG(work_floats >= nb * C, "ff_dwconv3_gate_pool: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dwconv3_gate_pool_kernel, dim3((unsigned)nb), dim3(256), (size_t)ppb * cv * 16, st, in, ldi, out, ldo, H, W, C,
                     w_tapmajor, bias, work);
  hipLaunchKernelGGL(pool_reduce_kernel, dim3((C + 15) / 16), dim3(256), 0, st, work, (int)nb, C, 1.0f / (float)P, pooled);
  FF_LAUNCH_CHECK("ff_dwconv3_gate_pool");
  return FF_OK;
}

// --------------------------------------------------------------------------------------------- flash gated FFN
struct NafFfnParams {
  const float* y; float* out;
  const float* gamma_ln; const float* beta_ln;
  const __bf16* w;        // [GT][ W4a_hi, W4a_lo, W4b_hi, W4b_lo (32 x KPAD each), W5_hi, W5_lo (C x 32 each, permuted cols) ]
  const float* b4;        // [2C]
  const float* b5;        // [C]
  const float* oscale;    // [C] (NAFBlock gamma)
  long long M;
  int ldy, ldo;
  float eps;
};

template <int KS, int NTERMS>
__global__ __launch_bounds__(512) void naf_ffn_kernel(NafFfnParams p) {
  constexpr int C = 16 * KS, GT = C / 32, NTO = C / 32;
  constexpr int W1SLOTS = 2 * KS + 1, W1ROWB = W1SLOTS * 16, W1PL = 32 * W1SLOTS;      // per plane (one 32-row tile)
  constexpr int W2ROWB = 80, W2PL = C * 5;
  constexpr int W1B = W1PL * 16, W2B = W2PL * 16;
  constexpr int W1PIECES = 4 * W1PL / 64, W2PIECES = 2 * W2PL / 64;
  constexpr int W1ELEMS = 32 * C, W2ELEMS = C * 32;
  constexpr int REC = 4 * W1ELEMS + 2 * W2ELEMS;
  constexpr int XLROWB = KS * 32 + 16;
  constexpr int N1 = (W1PIECES + 7) / 8, N2 = (W2PIECES + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* W1s = smem;                          // [4 planes: a_hi, a_lo, b_hi, b_lo][32][W1ROWB]
  unsigned char* W2s = smem + 4 * W1B;                // [2 planes][C][80]
  unsigned char* XLs = W2s + 2 * W2B;                 // [8 waves][32][XLROWB]
  float* B4s = reinterpret_cast<float*>(XLs + 8 * 32 * XLROWB);        // [2C]
  float* XSt = B4s + 2 * C;                           // x staging patches (KS == 4 only: the x_lo rows are too small to alias)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const long long tok0 = (long long)blockIdx.x * 256 + wid * 32;

  int off1[N1], off2[N2];
#pragma unroll
  for (int i = 0; i < N1; ++i) {
    int s = (wid + 8 * i) * 64 + lane;
    if (s >= 4 * W1PL) s = 4 * W1PL - 1;
    const int plane = s / W1PL, t = s - plane * W1PL, row = t / W1SLOTS;
    int q = t - row * W1SLOTS;
    if (q > 2 * KS - 1) q = 2 * KS - 1;
    off1[i] = plane * W1ELEMS + row * C + q * 8;
  }
#pragma unroll
  for (int i = 0; i < N2; ++i) {
    int s = (wid + 8 * i) * 64 + lane;
    if (s >= 2 * W2PL) s = 2 * W2PL - 1;
    const int plane = s / W2PL, t = s - plane * W2PL, row = t / 5;
    int q = t - row * 5;
    if (q > 3) q = 3;
    off2[i] = 4 * W1ELEMS + plane * W2ELEMS + row * 32 + q * 8;
  }
  auto dma_w1 = [&](int g) {
    const __bf16* rec = p.w + (long long)g * REC;
#pragma unroll
    for (int i = 0; i < N1; ++i)
      if (wid + 8 * i < W1PIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + off1[i]),
                                         (__attribute__((address_space(3))) void*)(W1s + (wid + 8 * i) * 1024), 16, 0, 0);
  };
  auto dma_w2 = [&](int g) {
    const __bf16* rec = p.w + (long long)g * REC;
#pragma unroll
    for (int i = 0; i < N2; ++i)
      if (wid + 8 * i < W2PIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + off2[i]),
                                         (__attribute__((address_space(3))) void*)(W2s + (wid + 8 * i) * 1024), 16, 0, 0);
  };
  dma_w1(0);
  dma_w2(0);
  for (int i = tid; i < 2 * C; i += 512) B4s[i] = p.b4[i];

  bf16x8 xh[KS];
  unsigned char* xl_row = XLs + (size_t)(wid * 32 + l31) * XLROWB + 16 * hh;
  {
    float v[KS][8];
    float* xs = (KS >= 8) ? reinterpret_cast<float*>(XLs + (size_t)wid * 32 * XLROWB) : XSt + wid * (32 * FF_XS_ROW);
    ff_wave_rows_to_frags<KS / 4>(p.y, p.ldy, tok0, p.M, C, xs, lane, v);
    float s = 0.f;
#pragma unroll
    for (int st = 0; st < KS; ++st)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[st][j];
    s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float qv = 0.f;
#pragma unroll
    for (int st = 0; st < KS; ++st)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[st][j] - mean; qv += d * d; }
    qv += __shfl_xor(qv, 32);
    const float rstd = 1.0f / sqrtf(qv / (float)C + p.eps);
#pragma unroll
    for (int st = 0; st < KS; ++st) {
      const int k0 = 16 * st + 8 * hh;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma_ln + k0), g1 = *reinterpret_cast<const f32x4*>(p.gamma_ln + k0 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta_ln + k0), b1 = *reinterpret_cast<const f32x4*>(p.beta_ln + k0 + 4);
      bf16x8 lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (v[st][j] - mean) * rstd * (j < 4 ? g0[j & 3] : g1[j & 3]) + (j < 4 ? b0[j & 3] : b1[j & 3]);
        const __bf16 h = (__bf16)f;
        xh[st][j] = h;
        lo[j] = (__bf16)(f - (float)h);
      }
      *reinterpret_cast<bf16x8*>(xl_row + 32 * st) = lo;
    }
  }

  f32x16 oacc[NTO];
#pragma unroll
  for (int n = 0; n < NTO; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[n][r] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int g = 0; g < GT; ++g) {
    // ---- two H^T tiles: channels [32g, 32g+32) and [C + 32g, C + 32g + 32) of conv4 ---------------------------------
    f32x16 ha, hb;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ch = 32 * g + (r & 3) + 8 * (r >> 2) + 4 * hh;
      ha[r] = B4s[ch];
      hb[r] = B4s[C + ch];
    }
    {
      const unsigned char* ap = W1s + l31 * W1ROWB + 16 * hh;
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        const bf16x8 a_h = *reinterpret_cast<const bf16x8*>(ap + 32 * st);
        const bf16x8 a_l = *reinterpret_cast<const bf16x8*>(ap + 32 * st + W1B);
        const bf16x8 b_h = *reinterpret_cast<const bf16x8*>(ap + 32 * st + 2 * W1B);
        const bf16x8 b_l = *reinterpret_cast<const bf16x8*>(ap + 32 * st + 3 * W1B);
        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(xl_row + 32 * st);
        if (NTERMS == 3) {
          ha = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, xl, ha, 0, 0, 0);
          hb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_h, xl, hb, 0, 0, 0);
          ha = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, xh[st], ha, 0, 0, 0);
          hb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_l, xh[st], hb, 0, 0, 0);
        }
        ha = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, xh[st], ha, 0, 0, 0);
        hb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_h, xh[st], hb, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // W5(g) pieces of this wave have landed
    __syncthreads();                                    // A: all waves done with the W4 image; W5(g) visible
    if (g + 1 < GT) dma_w1(g + 1);
    // ---- SimpleGate: product of the two tiles; registers 8s..8s+7 are the B fragment of k-step s ---------------------
    bf16x8 gh[2], gl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gt = ha[8 * s + j] * hb[8 * s + j];
        const __bf16 h = (__bf16)gt;
        gh[s][j] = h;
        gl[s][j] = (__bf16)(gt - (float)h);
      }
#pragma unroll
    for (int n = 0; n < NTO; ++n)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const unsigned char* ap = W2s + (n * 32 + l31) * W2ROWB + 32 * s + 16 * hh;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + W2B);
        if (NTERMS == 3) {
          oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gl[s], oacc[n], 0, 0, 0);
          oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gh[s], oacc[n], 0, 0, 0);
        }
        oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh[s], oacc[n], 0, 0, 0);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // W4(g+1) pieces landed
    __syncthreads();                                    // B: all waves done with the W5 image
    if (g + 1 < GT) dma_w2(g + 1);
  }

  // ---- epilogue: out = y + oscale[n] * (acc + b5[n]); residual loads first, transpose through LDS, coalesced stores ----
  float* tr = reinterpret_cast<float*>(smem) + wid * (32 * 33);
  float rv[NTO][16];
#pragma unroll
  for (int n = 0; n < NTO; ++n) {
    const int col = n * 32 + l31;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long long tk = tok0 + 2 * i + hh;
      rv[n][i] = p.y[(tk < p.M ? tk : 0) * p.ldy + col];
    }
  }
#pragma unroll
  for (int n = 0; n < NTO; ++n) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tr[l31 * 33 + (r & 3) + 8 * (r >> 2) + 4 * hh] = oacc[n][r];
    const int col = n * 32 + l31;
    const float sc = p.oscale[col], bb = p.b5[col];
    float ov[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ov[i] = (tr[(2 * i + hh) * 33 + l31] + bb) * sc + rv[n][i];
    float* op = p.out + (tok0 + hh) * p.ldo + col;
    if (tok0 + 32 <= p.M) {
#pragma unroll
      for (int i = 0; i < 16; ++i) op[(long long)(2 * i) * p.ldo] = ov[i];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (tok0 + 2 * i + hh < p.M) op[(long long)(2 * i) * p.ldo] = ov[i];
    }
  }
}

template <int KS, int NTERMS>
static int launch_naf_ffn(const NafFfnParams& p, hipStream_t st) {
  constexpr int C = 16 * KS;
  const size_t lds = (size_t)4 * 32 * (2 * KS + 1) * 16 + (size_t)2 * C * 5 * 16 + (size_t)8 * 32 * (KS * 32 + 16) + (size_t)2 * C * 4 +
                     (KS >= 8 ? 0 : (size_t)8 * 32 * FF_XS_ROW * 4);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&naf_ffn_kernel<KS, NTERMS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { ff_set_error("ff_naf_ffn: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; }
    attr_set = true;
  }
  const long long nblk = (p.M + 255) / 256;
  if (nblk >= (1LL << 31)) { ff_set_error("ff_naf_ffn: grid too large"); return FF_ERR_ARG; }
  hipLaunchKernelGGL((naf_ffn_kernel<KS, NTERMS>), dim3((unsigned)nblk), dim3(512), lds, st, p);
  FF_LAUNCH_CHECK("ff_naf_ffn");
  return FF_OK;
}

extern "C" int ff_naf_ffn(const float* y, int ldy, float* out, int ldo, long long M, int C, const float* gamma_ln,
                          const float* beta_ln, float eps, const void* w_tiles, const float* b4, const float* b5,
                          const float* out_scale, int nterms, void* stream) {
  FF_CHECK_ARG(nterms == 1 || nterms == 3, "ff_naf_ffn: nterms must be 1 or 3");
  FF_CHECK_ARG(y && out && gamma_ln && beta_ln && w_tiles && b4 && b5 && out_scale, "ff_naf_ffn: null pointer");
  FF_CHECK_ARG(M > 0 && (C == 64 || C == 128), "ff_naf_ffn: C must be 64 or 128");
  FF_CHECK_ARG(ldy >= C && ldy % 4 == 0 && ldo >= C && (((uintptr_t)y) & 15) == 0 && (((uintptr_t)w_tiles) & 15) == 0 &&
               (((uintptr_t)gamma_ln) & 15) == 0 && (((uintptr_t)beta_ln) & 15) == 0, "ff_naf_ffn: 16-byte alignment required");
  NafFfnParams p;
  p.y = y; p.out = out; p.gamma_ln = gamma_ln; p.beta_ln = beta_ln; p.w = (const __bf16*)w_tiles; p.b4 = b4; p.b5 = b5;
  p.oscale = out_scale; p.M = M; p.ldy = ldy; p.ldo = ldo; p.eps = eps;
  if (nterms == 3) return C == 64 ? launch_naf_ffn<4, 3>(p, (hipStream_t)stream) : launch_naf_ffn<8, 3>(p, (hipStream_t)stream);
  return C == 64 ? launch_naf_ffn<4, 1>(p, (hipStream_t)stream) : launch_naf_ffn<8, 1>(p, (hipStream_t)stream);
}

#include "naf_front.inc"

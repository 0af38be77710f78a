// GEMM on pre-split bf16 planes:  out[m][n] = res[m][n] + alpha * mul[n] * act( sum_k A[m][k] * B[n][k] + bias[n] )
// with A = a_hi + a_lo and B = b_hi + b_lo both ALREADY stored as bf16 planes [rows][Kp] (Kp % 32 == 0, zero padded):
// the weights by ff_split_bf16 at load time, the activations by their PRODUCER kernel (LayerNorm, the gated depth-wise conv,
// the SimpleGate product write hi/lo planes instead of fp32 -- same bytes).  Unlike conv_gemm_bf16.hip there is then no
// fp32 -> bf16 conversion in the K loop (it was repeated by every one of the N/128 workgroups that share a row block) and
// no register staging: both operand tiles stream global -> LDS by LDS-DMA (buffer_load ... lds) through a 3-stage ring, and
// the 8 waves (two per SIMD) run nothing but ds_read_b128 + v_mfma_f32_32x32x16_bf16 (bf16x3: hi*lo + lo*hi + hi*hi).
// Used for the 1x1 convolutions with K >= 256 (NAFNet levels 2-4, nafnet_arch.py:77,82,95,96) -- the GEMM-bound part of the path.
#include "ff_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct GemmPlParams {
  const __bf16* a_hi; const __bf16* a_lo; const __bf16* b_hi; const __bf16* b_lo;
  const float* bias; const float* mul; const float* res; float* out;
  int M, N, Kp, ldo, ldr, act;
  float alpha;
};

#define GP_BM 128
#define GP_BN 128
#define GP_ROWB 144                      // 64 B hi | 64 B lo | 16 B pad  (conflict-free ds_read_b128)
#define GP_SLOTS 9
#define GP_ROWS (GP_BM + GP_BN)
#define GP_STAGEB (GP_ROWS * GP_ROWB)    // 36 864 B = 36 DMA pieces
#define GP_PIECES (GP_STAGEB / 1024)
#define GP_NSTAGE 3

template <int NTERMS>
__global__ __launch_bounds__(512) void gemm_planes_kernel(GemmPlParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wr = wid >> 2, wc = wid & 3;                        // 2 (M) x 4 (N) waves, 64 x 32 each
  const int mtiles = (p.M + GP_BM - 1) / GP_BM, ntiles = (p.N + GP_BN - 1) / GP_BN;
  const int L = ff_xcd_remap(blockIdx.x, mtiles * ntiles);
  const int m0 = (L / ntiles) * GP_BM, n0 = (L % ntiles) * GP_BN;

  // ---- DMA bookkeeping: this wave moves pieces wid, wid + 8, ... of every stage; per lane the global source of its slot ----
  const __bf16* src[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    int s = (wid + 8 * i) * 64 + lane;
    if (s >= GP_ROWS * GP_SLOTS) s = GP_ROWS * GP_SLOTS - 1;
    const int row = s / GP_SLOTS;
    int q = s - row * GP_SLOTS;
    if (q > 7) q = 7;                                            // the pad slot re-reads its neighbour (never consumed)
    const bool lo = q >= 4;
    if (row < GP_BM) {
      int m = m0 + row;
      if (m >= p.M) m = p.M - 1;
      src[i] = (lo ? p.a_lo : p.a_hi) + (long long)m * p.Kp + 8 * (q & 3);
    } else {
      int n = n0 + row - GP_BM;
      if (n >= p.N) n = p.N - 1;
      src[i] = (lo ? p.b_lo : p.b_hi) + (long long)n * p.Kp + 8 * (q & 3);
    }
  }
  auto dma = [&](int chunk, int stage) {
    unsigned char* dst = smem + stage * GP_STAGEB;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (wid + 8 * i < GP_PIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + chunk * 32),
                                         (__attribute__((address_space(3))) void*)(dst + (wid + 8 * i) * 1024), 16, 0, 0);
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int nchunks = p.Kp / 32;
  dma(0, 0);
  if (nchunks > 1) dma(1, 1);
  for (int c = 0; c < nchunks; ++c) {
    const int stage = c % GP_NSTAGE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // own pieces of chunk c (and c+1) have landed
    __builtin_amdgcn_s_barrier();                               // ... everyone's; every wave is past its reads of stage (c+2)%3
    if (c + 2 < nchunks) dma(c + 2, (c + 2) % GP_NSTAGE);
    const unsigned char* As = smem + stage * GP_STAGEB + (wr * 64 + l31) * GP_ROWB + 16 * hh;
    const unsigned char* Bs = smem + stage * GP_STAGEB + (GP_BM + wc * 32 + l31) * GP_ROWB + 16 * hh;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[2], al[2], bh, bl;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * GP_ROWB + 32 * s);
        if (NTERMS == 3) al[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * GP_ROWB + 32 * s + 64);
      }
      bh = *reinterpret_cast<const bf16x8*>(Bs + 32 * s);
      if (NTERMS == 3) bl = *reinterpret_cast<const bf16x8*>(Bs + 32 * s + 64);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (NTERMS == 3) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i], 0, 0, 0);
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                  // the ring is idle: reuse it as the transpose patches

  // ---- epilogue: acc[i] holds rows (r&3)+8(r>>2)+4hh of M-subtile i (A rows on the registers), column n = lane --------------
  // (D = A_tile . B_tile^T with A as the MFMA A operand: lane = n, registers = m.)  Transposed through a wave-private patch
  // so a lane owns four consecutive channels of one row: float4 residual loads and stores.
  auto epilogue = [&](auto ACTC) {
    constexpr int ACT = decltype(ACTC)::value;
    float* tr = reinterpret_cast<float*>(smem) + wid * (32 * 36);
    const int tq = lane >> 3, q4 = 4 * (lane & 7);
    const int n = n0 + wc * 32 + l31;
    const int nc = n < p.N ? n : 0;
    const float bv = p.bias ? p.bias[nc] : 0.f;
    const float mv = (p.mul ? p.mul[nc] : 1.f) * p.alpha;
    const int nq = n0 + wc * 32 + q4;
    const bool nqok = nq < p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mb = m0 + wr * 64 + i * 32;
      f32x4 rq[4];
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int m = mb + tq + 8 * k;
        ok[k] = nqok && m < p.M;
        f32x4 r0 = {0.f, 0.f, 0.f, 0.f};
        if (p.res) { const f32x4 u = *reinterpret_cast<const f32x4*>(p.res + (ok[k] ? (long long)m * p.ldr + nq : 0)); r0 = ok[k] ? u : r0; }
        rq[k] = r0;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tr[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + l31] = ff_act_c<ACT, true>(acc[i][r] + bv) * mv;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 ov = *reinterpret_cast<const f32x4*>(tr + (tq + 8 * k) * 36 + q4) + rq[k];
        if (ok[k]) *reinterpret_cast<f32x4*>(p.out + (long long)(mb + tq + 8 * k) * p.ldo + nq) = ov;
      }
    }
  };
  FF_DISPATCH_ACT(p.act, epilogue)
}

extern "C" int ff_gemm_planes(const void* a_hi, const void* a_lo, const void* b_hi, const void* b_lo, int M, int N, int Kp,
                              const float* bias, const float* mul, const float* res, int ldr, float* out, int ldo, int act,
                              float alpha, int nterms, void* stream) {
  FF_CHECK_ARG(a_hi && b_hi && out && M > 0 && N > 0 && Kp > 0 && Kp % 32 == 0, "ff_gemm_planes: bad args (Kp must be a multiple of 32)");
  FF_CHECK_ARG(nterms == 1 || (nterms == 3 && a_lo && b_lo), "ff_gemm_planes: nterms must be 1, or 3 with both lo planes");
  FF_CHECK_ARG(N % 4 == 0 && ldo % 4 == 0 && ldo >= N && (((uintptr_t)out) & 15) == 0, "ff_gemm_planes: out rows must be 16-byte aligned, N %% 4 == 0");
  FF_CHECK_ARG(!res || (ldr % 4 == 0 && ldr >= N && (((uintptr_t)res) & 15) == 0), "ff_gemm_planes: res rows must be 16-byte aligned");
  FF_CHECK_ARG(((((uintptr_t)a_hi) | ((uintptr_t)b_hi) | ((uintptr_t)a_lo) | ((uintptr_t)b_lo)) & 15) == 0, "ff_gemm_planes: planes must be 16-byte aligned");
  GemmPlParams p;
  p.a_hi = (const __bf16*)a_hi; p.a_lo = (const __bf16*)(a_lo ? a_lo : a_hi); p.b_hi = (const __bf16*)b_hi; p.b_lo = (const __bf16*)(b_lo ? b_lo : b_hi);
  p.bias = bias; p.mul = mul; p.res = res; p.out = out; p.M = M; p.N = N; p.Kp = Kp; p.ldo = ldo; p.ldr = ldr; p.act = act; p.alpha = alpha;
  const long long nblk = (long long)ff_cdiv(M, GP_BM) * ff_cdiv(N, GP_BN);
  FF_CHECK_ARG(nblk < (1LL << 31), "ff_gemm_planes: grid too large");
  const size_t lds = (size_t)GP_NSTAGE * GP_STAGEB;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_planes_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_planes_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { ff_set_error("ff_gemm_planes: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; }
    attr_set = true;
  }
  if (nterms == 3) hipLaunchKernelGGL(gemm_planes_kernel<3>, dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gemm_planes_kernel<1>, dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, p);
  FF_LAUNCH_CHECK("ff_gemm_planes");
  return FF_OK;
}

// fp32 rows [M][ld] (K channels) -> bf16 planes hi / lo [M][Kp], zero padded: the stand-alone producer (activations whose
// producer kernel cannot emit planes itself)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, int ld, long long M, int K, int Kp,
                                                         __bf16* __restrict__ hi, __bf16* __restrict__ lo) {
  const int q = Kp >> 2;
  const long long total = M * q;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / q;
    const int k = (int)(i - m * q) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k < K) v = *reinterpret_cast<const f32x4*>(x + m * ld + k);
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float f = (k + e < K) ? v[e] : 0.f;
      h[e] = (__bf16)f;
      l[e] = (__bf16)(f - (float)h[e]);
    }
    *reinterpret_cast<bf16x4*>(hi + m * Kp + k) = h;
    if (lo) *reinterpret_cast<bf16x4*>(lo + m * Kp + k) = l;
  }
}

extern "C" int ff_split_rows(const float* x, int ld, long long M, int K, int Kp, void* hi, void* lo, void* stream) {
  FF_CHECK_ARG(x && hi && M > 0 && K > 0 && K % 4 == 0 && Kp % 32 == 0 && Kp >= K && ld % 4 == 0 && ld >= K && (((uintptr_t)x) & 15) == 0,
               "ff_split_rows: rows must be 16-byte aligned, K %% 4 == 0, Kp a multiple of 32");
  long long nb = (M * (Kp / 4) + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ld, M, K, Kp, (__bf16*)hi, (__bf16*)lo);
  FF_LAUNCH_CHECK("ff_split_rows");
  return FF_OK;
}

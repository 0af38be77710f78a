// Implicit-GEMM convolution / GEMM on the CDNA4 matrix cores, fp32 in / fp32 accumulate
// (v_mfma_f32_32x32x2_f32: exact-f32 fmaf chain, 64 FLOP/clk/SIMD, guide section 3).
//
//   out[m][n] = res[m][n] + alpha * mul[n] * act( sum_k A[m][k] * Wt[n][k] + bias[n] )
//
// m = output pixel (b, oy, ox) of an NHWC tensor, n = output channel, k = (ky, kx, ci).
// A is gathered on the fly from the NHWC input (zero padding), so every dense contraction of
// the path -- nn.Linear, 1x1 conv, 3x3 conv, 2x2 stride-2 conv -- is this one kernel
// (reference call sites: hat_arch.py:67-69,89-92,172,194,608; dat_arch.py:163-168,501,559,792;
// nafnet_arch.py:77-96,174,184; hierarchical_fusion.py:96-127; enhanced_fusion.py:266-290).
//
// Tiling: workgroup = 4 waves (256 threads), BM x BN output tile, BK = 16.  LDS rows are
// k-contiguous with a 4-float pad (LDK = 20): the lane's operand for 8 consecutive k-steps is two
// ds_read_b128 (lane half h owns k = 8h..8h+7 of the chunk -- the k order inside a chunk is free as
// long as A and B agree), and rows r, r+1, ... start 20 dwords apart, which puts the 16 rows of a
// ds_read_b128 lane group on 16 distinct 4-bank slots (conflict free).  Global->LDS goes through
// registers (the im2col gather cannot be expressed as a lane-linear LDS-DMA image), software
// pipelined one chunk ahead with two LDS buffers and one barrier per chunk.
#include "ff_common.h"

struct ConvParams {
  const float* in;
  const float* w;
  const float* bias;
  const float* mul;
  const float* res;
  float* out;
  int B, H, W, Cin, ldi;
  int Ho, Wo, Cout, ldo, ldr;
  int KH, KW, sy, sx, py, px;
  int K, M;
  int act;
  float alpha;
  int shuffle;  // 0 or 2: PixelShuffle(2) folded into the store
};

#define LDK 20
#define BK 16

template <int BM, int BN, int WM, int WN, bool VEC4>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int AR = BM / 64;                    // A rows staged per thread
  constexpr int BR = (BN + 63) / 64;             // B rows staged per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                              // [2][BM][LDK]
  float* Bs = smem + 2 * BM * LDK;               // [2][BN][LDK]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wr = wid / WN, wc = wid % WN;

  const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.Cout + BN - 1) / BN;
  const int L = ff_xcd_remap(blockIdx.x, mtiles * ntiles);
  const int m0 = (L / ntiles) * BM, n0 = (L % ntiles) * BN;

  // --- per-thread staging rows --------------------------------------------------------------
  const int srow = tid >> 2, kq = tid & 3;
  int a_b[AR], a_iy[AR], a_ix[AR];
  bool a_ok[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + srow + 64 * i;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    const int ox = mm % p.Wo, t2 = mm / p.Wo;
    const int oy = t2 % p.Ho;
    a_b[i] = t2 / p.Ho;
    a_iy[i] = oy * p.sy - p.py;
    a_ix[i] = ox * p.sx - p.px;
  }
  const bool is1x1 = (p.KH == 1 && p.KW == 1);

  f32x4 ra[AR], rb[BR];

  auto load_chunk = [&](int k0) {
    if (VEC4) {
      const int k = k0 + 4 * kq;
      const bool kok = k < p.K;
      int ky = 0, kx = 0, ci = k;
      if (!is1x1 && kok) {
        const int tap = k / p.Cin;
        ci = k - tap * p.Cin;
        ky = tap / p.KW;
        kx = tap - ky * p.KW;
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kok && a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          v = *reinterpret_cast<const f32x4*>(p.in + ((long long)(a_b[i] * p.H + iy) * p.W + ix) * p.ldi + ci);
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int r = srow + 64 * i;
        const int n = n0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < BN && kok && n < p.Cout) v = *reinterpret_cast<const f32x4*>(p.w + (long long)n * p.K + k);
        rb[i] = v;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + 4 * kq + e;
        const bool kok = k < p.K;
        int ky = 0, kx = 0, ci = k;
        if (!is1x1 && kok) {
          const int tap = k / p.Cin;
          ci = k - tap * p.Cin;
          ky = tap / p.KW;
          kx = tap - ky * p.KW;
        }
#pragma unroll
        for (int i = 0; i < AR; ++i) {
          const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
          float v = 0.f;
          if (kok && a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
            v = p.in[((long long)(a_b[i] * p.H + iy) * p.W + ix) * p.ldi + ci];
          ra[i][e] = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
          const int r = srow + 64 * i;
          const int n = n0 + r;
          float v = 0.f;
          if (r < BN && kok && n < p.Cout) v = p.w[(long long)n * p.K + k];
          rb[i][e] = v;
        }
      }
    }
  };

  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i)
      *reinterpret_cast<f32x4*>(As + (buf * BM + srow + 64 * i) * LDK + 4 * kq) = ra[i];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int r = srow + 64 * i;
      if (r < BN) *reinterpret_cast<f32x4*>(Bs + (buf * BN + r) * LDK + 4 * kq) = rb[i];
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = (p.K + BK - 1) / BK;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) load_chunk((c + 1) * BK);
    f32x4 fa[MI][2], fb[NI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const float* ap = As + (buf * BM + wr * TM + i * 32 + l31) * LDK + 8 * hh;
      fa[i][0] = *reinterpret_cast<const f32x4*>(ap);
      fa[i][1] = *reinterpret_cast<const f32x4*>(ap + 4);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const float* bp = Bs + (buf * BN + wc * TN + j * 32 + l31) * LDK + 8 * hh;
      fb[j][0] = *reinterpret_cast<const f32x4*>(bp);
      fb[j][1] = *reinterpret_cast<const f32x4*>(bp + 4);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s >> 2][s & 3], fb[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
    if (c + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // --- epilogue: lane = output channel, registers = output pixels -----------------------------
  auto epilogue = [&](auto ACTC) {
    constexpr int ACT = decltype(ACTC)::value;
    const bool has_res = p.res != nullptr;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wc * TN + j * 32 + l31;
      if (n >= p.Cout) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
      const float mv = (p.mul ? p.mul[n] : 1.f) * p.alpha;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        // indices first, then ALL residual loads, then the stores: `out` may alias `res` as far as the compiler knows,
        // so a residual load placed after a store is serialised behind it (16 dependent L2 round trips per tile)
        long long oidx[16];
        float rv[16];
        const int mb = m0 + wr * TM + i * 32 + 4 * hh;
        if (p.shuffle == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            const int mm = m < p.M ? m : 0;
            const int ox = mm % p.Wo, t2 = mm / p.Wo;
            const int oy = t2 % p.Ho, b = t2 / p.Ho;
            const int co = n >> 2, dy = (n >> 1) & 1, dx = n & 1;
            const long long pix = ((long long)(b * 2 * p.Ho + 2 * oy + dy) * (2 * p.Wo) + 2 * ox + dx);
            oidx[r] = pix * p.ldo + co;
            rv[r] = has_res ? p.res[pix * p.ldr + co] : 0.f;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            const int mm = m < p.M ? m : 0;
            oidx[r] = (long long)mm * p.ldo + n;
            rv[r] = has_res ? p.res[(long long)mm * p.ldr + n] : 0.f;
          }
        }
        if (mb + 28 < p.M) {                      // whole 32-row tile in range: no per-store exec masking
#pragma unroll
          for (int r = 0; r < 16; ++r) p.out[oidx[r]] = ff_act_c<ACT, false>(acc[i][j][r] + bv) * mv + rv[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            if (m < p.M) p.out[oidx[r]] = ff_act_c<ACT, false>(acc[i][j][r] + bv) * mv + rv[r];
          }
        }
      }
    }
  };
  FF_DISPATCH_ACT(p.act, epilogue)
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const ConvParams& p, bool vec4, hipStream_t st) {
  const int mt = ff_cdiv(p.M, BM), nt = ff_cdiv(p.Cout, BN);
  const size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  dim3 grid((unsigned)(mt * nt)), block(256);
  if (vec4)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, true>), grid, block, lds, st, p);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false>), grid, block, lds, st, p);
  FF_LAUNCH_CHECK("ff_conv2d");
  return FF_OK;
}

extern "C" int ff_conv2d(const float* in, const float* w, const float* bias, const float* mul, const float* res,
                         float* out, int B, int H, int W, int Cin, int ldi, int Ho, int Wo, int Cout, int ldo,
                         int ldr, int KH, int KW, int sy, int sx, int py, int px, int act, float alpha, int shuffle,
                         int tile_hint, void* stream) {
  FF_CHECK_ARG(in && w && out, "ff_conv2d: null pointer");
  FF_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Ho > 0 && Wo > 0, "ff_conv2d: bad dims");
  FF_CHECK_ARG(ldi >= Cin, "ff_conv2d: ldi %d < Cin %d", ldi, Cin);
  FF_CHECK_ARG(KH > 0 && KW > 0 && sy > 0 && sx > 0 && py >= 0 && px >= 0, "ff_conv2d: bad kernel geometry");
  FF_CHECK_ARG(shuffle == 0 || shuffle == 2, "ff_conv2d: shuffle must be 0 or 2");
  FF_CHECK_ARG(shuffle == 0 || Cout % 4 == 0, "ff_conv2d: shuffle needs Cout %% 4 == 0");
  FF_CHECK_ARG(ldo >= (shuffle ? Cout / 4 : Cout), "ff_conv2d: ldo too small");
  FF_CHECK_ARG(!res || ldr >= (shuffle ? Cout / 4 : Cout), "ff_conv2d: ldr too small");
  FF_CHECK_ARG((long long)B * Ho * Wo < (1LL << 31), "ff_conv2d: M overflows int");
  ConvParams p;
  p.in = in; p.w = w; p.bias = bias; p.mul = mul; p.res = res; p.out = out;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.ldi = ldi;
  p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.ldo = ldo; p.ldr = ldr;
  p.KH = KH; p.KW = KW; p.sy = sy; p.sx = sx; p.py = py; p.px = px;
  p.K = KH * KW * Cin; p.M = B * Ho * Wo;
  p.act = act; p.alpha = alpha; p.shuffle = shuffle;
  const bool vec4 = (Cin % 4 == 0) && (ldi % 4 == 0) && (((uintptr_t)in & 15) == 0) && (((uintptr_t)w & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  int cfg = tile_hint;
  if (cfg <= 0) {
    if (Cout <= 32) cfg = 3;
    else if (Cout % 128 == 0 && p.M >= 128 * 512) cfg = 1;
    else cfg = 2;
  }
  switch (cfg) {
    case 1: return launch_cfg<128, 128, 2, 2>(p, vec4, st);
    case 2: return launch_cfg<128, 64, 2, 2>(p, vec4, st);
    case 3: return launch_cfg<256, 32, 4, 1>(p, vec4, st);
    default: ff_set_error("ff_conv2d: bad tile_hint %d", tile_hint); return FF_ERR_ARG;
  }
}

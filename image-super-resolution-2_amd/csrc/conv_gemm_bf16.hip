// Implicit-GEMM convolution / GEMM on the bf16 matrix cores with fp32-grade accuracy by operand
// splitting ("bf16x3"):  a = a_hi + a_lo (+ O(2^-16 |a|)),  both halves bf16, so
//     a*w  ~=  a_hi*w_hi + a_lo*w_hi + a_hi*w_lo          (the dropped a_lo*w_lo term is O(2^-16))
// Each term is one v_mfma_f32_32x32x16_bf16 into the SAME fp32 accumulator: 3 MFMAs at 16x the fp32
// MFMA rate = 5.3x the fp32-MFMA ceiling of conv_gemm.hip at the same results to ~1e-5 relative.
// NTERMS = 1 (plain bf16) and 2 (fp32-accurate activations x bf16 weights) are the cheaper modes.
//
// Same contract / epilogue as conv_gemm.hip.  Differences:
//   - weights arrive pre-split by ff_split_bf16 as two bf16 planes w_hi/w_lo [Cout][Kp], Kp = K rounded up
//     to 32 (zero filled), so the B tile is a straight 16-byte copy with no K tail;
//   - activations are split on the fly in the staging path (v_cvt_pk_bf16_f32), once per element per
//     workgroup, and stored to LDS as [row][hi k0..31 | lo k0..31] (128 B + 16 B pad = 36 dwords:
//     36 = 4*9 puts the 16 rows of a ds_read_b128 lane group on 16 distinct 4-bank slots);
//   - BK = 32: two k-steps of the 32x32x16 MFMA per chunk; a lane's operand is one ds_read_b128
//     (8 consecutive k of its row) per plane and k-step;
//   - K is laid out per filter tap with the tap's Cin padded to Cp = ceil32(Cin) when Cin >= 32 ("TAP" mode:
//     a chunk never straddles taps, so (tap, ci0) are wave-uniform scalars and the per-thread gather needs no
//     division); small-Cin convolutions (3, 9, 27 ... input channels) keep the flat k = (tap, ci) order;
//   - tiles: 256x192 with 8 waves (64x96 per wave, 36 MFMAs per chunk and wave) for the 180/360/540/720-wide
//     transformer layers (65536 tokens -> exactly 256 workgroups per 192 columns: one per CU, no tail),
//     128x128 / 128x64 / 256x32 with 4 waves elsewhere.
#include "ff_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct ConvBfParams {
  const float* in;
  const __bf16* w_hi;
  const __bf16* w_lo;
  const float* bias;
  const float* mul;
  const float* res;
  const float* kmul;   // optional [Cin] scale of the INPUT channels (1x1 only): out = W (kmul * a) -- NAFNet's conv3(x * sca)
  float* out;
  int B, H, W, Cin, ldi;
  int Ho, Wo, Cout, ldo, ldr;
  int KH, KW, sy, sx, py, px;
  int K, Kp, M;
  int Cp;          // TAP mode: per-tap padded Cin (multiple of 32); 0 = flat k order
  int act;
  float alpha;
  int shuffle;
  int vec_out;     // 16-byte epilogue: no shuffle, Cout % 4 == 0, 16-byte aligned out / res rows
  int gate;        // shuffle == 1: SimpleGate in the epilogue, out[m][j] = y[m][2j] * y[m][2j+1] (Cout / 2 channels out)
};

// BKB = k-values per chunk (one barrier per chunk).  32: the original form.  64 (r2): twice the MFMAs between barriers -- the
// 8-wave 128-row tiles issue only 12 MFMAs per wave and chunk at 32, and the barrier + staging round trip per chunk, not the
// MFMA pipe, set their time (SQ counters, tools/gemm_prof.py: MFMA pipe 28 % busy on the 1024 -> 2048 NAFNet GEMM).
//   LDS row = [BKB hi | BKB lo | 16 B pad]: 144 B (36 dwords) or 272 B (68 dwords), both = 4 (mod 32) dwords: conflict-free b128 reads.
template <int BM, int BN, int WM, int WN, bool VEC4, int NTERMS, int BKB>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm_bf16_kernel(ConvBfParams p) {
  constexpr int ROWB = BKB * 4 + 16, PLB = BKB * 2;   // bytes per LDS row / per plane of a row
  constexpr int KQ = BKB / 4, QB = BKB / 8;       // 4-float quads / 16-byte bf16 items per row and plane
  constexpr int NT = WM * WN * 64;                // threads
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int RPP = NT / KQ;                    // A rows staged per pass
  constexpr int AQ = BM / RPP;                    // A quads (4 k-values) staged per thread
  constexpr int BI = (BN * 2 * QB + NT - 1) / NT; // B 16-byte items staged per thread (hi+lo planes)
  static_assert(BM % RPP == 0, "A staging must tile the rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                       // [2][BM][ROWB]
  unsigned char* Bs = smem + 2 * BM * ROWB;       // [2][BN][ROWB]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wr = wid / WN, wc = wid % WN;

  const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.Cout + BN - 1) / BN;
  const int L = ff_xcd_remap(blockIdx.x, mtiles * ntiles);
  const int m0 = (L / ntiles) * BM, n0 = (L % ntiles) * BN;

  const int srow = tid / KQ, kq = tid % KQ;       // A staging: row srow + RPP*i, k-quad kq
  int a_b[AQ], a_iy[AQ], a_ix[AQ];
  bool a_ok[AQ];
#pragma unroll
  for (int i = 0; i < AQ; ++i) {
    const int m = m0 + srow + RPP * i;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    const int ox = mm % p.Wo, t2 = mm / p.Wo;
    const int oy = t2 % p.Ho;
    a_b[i] = t2 / p.Ho;
    a_iy[i] = oy * p.sy - p.py;
    a_ix[i] = ox * p.sx - p.px;
  }
  const bool is1x1 = (p.KH == 1 && p.KW == 1);

  f32x4 ra[AQ];
  uint4 rb[BI];

  auto load_chunk = [&](int k0) {
    if (VEC4) {
      int ky = 0, kx = 0, ci;
      bool kok;
      if (p.Cp > 0) {                               // TAP mode: tap / ci0 are uniform (scalar) per chunk
        const int tap = k0 / p.Cp;
        ci = k0 - tap * p.Cp + 4 * kq;
        kok = ci < p.Cin;
        ky = tap / p.KW;
        kx = tap - ky * p.KW;
      } else {
        const int k = k0 + 4 * kq;
        kok = k < p.K;
        ci = k;
        if (!is1x1 && kok) {
          const int tap = k / p.Cin;
          ci = k - tap * p.Cin;
          ky = tap / p.KW;
          kx = tap - ky * p.KW;
        }
      }
#pragma unroll
      for (int i = 0; i < AQ; ++i) {
        const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kok && a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          v = *reinterpret_cast<const f32x4*>(p.in + ((long long)(a_b[i] * p.H + iy) * p.W + ix) * p.ldi + ci);
        ra[i] = v;
      }
      if (p.kmul && kok) {                            // (1x1: ci is the input channel; the launcher checks Cin % 4 == 0)
        const f32x4 km = *reinterpret_cast<const f32x4*>(p.kmul + ci);
#pragma unroll
        for (int i = 0; i < AQ; ++i) ra[i] *= km;
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
        for (int i = 0; i < AQ; ++i) {
          const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
          float v = 0.f;
          if (kok && a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
            v = p.in[((long long)(a_b[i] * p.H + iy) * p.W + ix) * p.ldi + ci];
          ra[i][e] = v;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      const int id = tid + NT * j;                 // [plane][row][q]
      const int plane = id / (BN * QB), rem = id % (BN * QB);
      const int r = rem / QB, q = rem % QB;
      uint4 v = {0u, 0u, 0u, 0u};
      if (id < BN * 2 * QB && (NTERMS == 3 || plane == 0) && n0 + r < p.Cout) {
        const __bf16* src = (plane ? p.w_lo : p.w_hi) + (long long)(n0 + r) * p.Kp + k0 + 8 * q;
        v = *reinterpret_cast<const uint4*>(src);
      }
      rb[j] = v;
    }
  };

  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AQ; ++i) {
      bf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float f = ra[i][e];
        const __bf16 h = (__bf16)f;
        hi[e] = h;
        lo[e] = (__bf16)(f - (float)h);
      }
      unsigned char* dst = As + (size_t)(buf * BM + srow + RPP * i) * ROWB + kq * 8;
      *reinterpret_cast<bf16x4*>(dst) = hi;
      if (NTERMS >= 2) *reinterpret_cast<bf16x4*>(dst + PLB) = lo;
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      const int id = tid + NT * j;
      const int plane = id / (BN * QB), rem = id % (BN * QB);
      const int r = rem / QB, q = rem % QB;
      if (id < BN * 2 * QB && (NTERMS == 3 || plane == 0))
        *reinterpret_cast<uint4*>(Bs + (size_t)(buf * BN + r) * ROWB + plane * PLB + q * 16) = rb[j];
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = p.Kp / BKB;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) load_chunk((c + 1) * BKB);
#pragma unroll
    for (int s = 0; s < BKB / 16; ++s) {
      bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const unsigned char* ap = As + (size_t)(buf * BM + wr * TM + i * 32 + l31) * ROWB + 32 * s + 16 * hh;
        ah[i] = *reinterpret_cast<const bf16x8*>(ap);
        if (NTERMS >= 2) al[i] = *reinterpret_cast<const bf16x8*>(ap + PLB);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const unsigned char* bp = Bs + (size_t)(buf * BN + wc * TN + j * 32 + l31) * ROWB + 32 * s + 16 * hh;
        bh[j] = *reinterpret_cast<const bf16x8*>(bp);
        if (NTERMS == 3) bl[j] = *reinterpret_cast<const bf16x8*>(bp + PLB);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if (NTERMS == 3) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          if (NTERMS >= 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    if (c + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  auto epilogue = [&](auto ACTC) {
    constexpr int ACT = decltype(ACTC)::value;
    const bool has_res = p.res != nullptr;
    if (p.gate) {
      // SimpleGate (nafnet_arch.py:28-31, 100-104) on interleaved columns: the caller orders the weight rows so that the two halves of
      // the reference's chunk(2, dim=1) alternate (2j <- j, 2j+1 <- j + C); a lane then multiplies adjacent columns of the
      // transposed 32 x 32 tile and stores four consecutive product channels: the 2C-wide tensor never reaches memory.
      float* tr = reinterpret_cast<float*>(smem) + wid * (32 * 36);
      const int grow = lane >> 2, seg = lane & 3;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wc * TN + j * 32 + l31;
        const float bv = (p.bias && n < p.Cout) ? p.bias[n] : 0.f;
        const int nq = (n0 + wc * TN + j * 32) / 2 + 4 * seg;      // first of this lane's four output channels
        const bool nqok = nq < p.Cout / 2;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int mb = m0 + wr * TM + i * 32;
#pragma unroll
          for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + l31] = acc[i][j][r] + bv;
#pragma unroll
          for (int pass = 0; pass < 2; ++pass) {
            const int row = grow + 16 * pass;
            const f32x4 a = *reinterpret_cast<const f32x4*>(tr + row * 36 + 8 * seg);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(tr + row * 36 + 8 * seg + 4);
            const f32x4 o = {a[0] * a[1], a[2] * a[3], b2[0] * b2[1], b2[2] * b2[3]};
            const int m = mb + row;
            if (nqok && m < p.M) *reinterpret_cast<f32x4*>(p.out + (long long)m * p.ldo + nq) = o;
          }
        }
      }
      return;
    }
    if (p.vec_out) {
      // 16-byte form: each 32(m) x 32(n) accumulator tile goes through a wave-private LDS patch (the staging buffers are
      // idle now) so a lane owns four consecutive channels of one row: float4 residual loads and stores, 8 whole
      // 128-byte row segments per instruction instead of 2 with dword accesses.
      float* tr = reinterpret_cast<float*>(smem) + wid * (32 * 36);
      const int tq = lane >> 3, q4 = 4 * (lane & 7);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wc * TN + j * 32 + l31;
        const int nc = n < p.Cout ? n : 0;
        const float bv = p.bias ? p.bias[nc] : 0.f;
        const float mv = (p.mul ? p.mul[nc] : 1.f) * p.alpha;
        const int nq = n0 + wc * TN + j * 32 + q4;
        const bool nqok = nq < p.Cout;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int mb = m0 + wr * TM + i * 32;
          f32x4 rq[4];
          long long oo[4];
          bool ok[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int m = mb + tq + 8 * k;
            ok[k] = nqok && m < p.M;
            oo[k] = ok[k] ? (long long)m * p.ldo + nq : 0;
            f32x4 r0 = {0.f, 0.f, 0.f, 0.f};
            if (has_res) { const f32x4 u = *reinterpret_cast<const f32x4*>(p.res + (ok[k] ? (long long)m * p.ldr + nq : 0)); r0 = ok[k] ? u : r0; }
            rq[k] = r0;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r)
            tr[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + l31] = ff_act_c<ACT, true>(acc[i][j][r] + bv) * mv;
          f32x4 ov[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) ov[k] = *reinterpret_cast<const f32x4*>(tr + (tq + 8 * k) * 36 + q4) + rq[k];
          if (nqok && mb + 32 <= p.M) {
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(p.out + oo[k]) = ov[k];
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (ok[k]) *reinterpret_cast<f32x4*>(p.out + oo[k]) = ov[k];
          }
        }
      }
      return;
    }
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
          for (int r = 0; r < 16; ++r) p.out[oidx[r]] = ff_act_c<ACT, true>(acc[i][j][r] + bv) * mv + rv[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            if (m < p.M) p.out[oidx[r]] = ff_act_c<ACT, true>(acc[i][j][r] + bv) * mv + rv[r];
          }
        }
      }
    }
  };
  FF_DISPATCH_ACT(p.act, epilogue)
}

template <int BM, int BN, int WM, int WN, int NT, int BKB = 32>
static int launch_bf(const ConvBfParams& p, bool vec4, hipStream_t st) {
  const int mt = ff_cdiv(p.M, BM), nt = ff_cdiv(p.Cout, BN);
  constexpr size_t lds = (size_t)2 * (BM + BN) * (BKB * 4 + 16);
  static_assert(lds <= 160 * 1024 && lds >= (size_t)WM * WN * 32 * 36 * 4, "LDS budget (staging ring; the epilogue patch reuses it)");
  dim3 grid((unsigned)(mt * nt)), block(WM * WN * 64);
  if (lds > 64 * 1024) {
    static bool attr_set[2] = {false, false};
    if (!attr_set[vec4 ? 1 : 0]) {
      hipError_t e = vec4 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_bf16_kernel<BM, BN, WM, WN, true, NT, BKB>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                          : hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_bf16_kernel<BM, BN, WM, WN, false, NT, BKB>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { ff_set_error("ff_conv2d_bf16s: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return FF_ERR_LAUNCH; }
      attr_set[vec4 ? 1 : 0] = true;
    }
  }
  if (vec4)
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<BM, BN, WM, WN, true, NT, BKB>), grid, block, lds, st, p);
  else
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<BM, BN, WM, WN, false, NT, BKB>), grid, block, lds, st, p);
  FF_LAUNCH_CHECK("ff_conv2d_bf16s");
  return FF_OK;
}

template <int NT>
static int dispatch_bf(const ConvBfParams& p, bool vec4, int cfg, hipStream_t st) {
  switch (cfg) {
    case 1: return launch_bf<128, 128, 2, 2, NT>(p, vec4, st);
    case 2: return launch_bf<128, 64, 2, 2, NT>(p, vec4, st);
    case 3: return launch_bf<256, 32, 4, 1, NT>(p, vec4, st);
    case 4: return launch_bf<256, 192, 4, 2, NT>(p, vec4, st);
    case 5: return launch_bf<128, 128, 2, 4, NT>(p, vec4, st);      // 8 waves (two per SIMD), 64x32 per wave
    case 6: return launch_bf<256, 128, 4, 2, NT>(p, vec4, st);      // 8 waves, 64x64 per wave
    case 7: return launch_bf<128, 64, 4, 2, NT>(p, vec4, st);       // 8 waves, 32x32 per wave
    // 64-deep chunks (Kp and the per-tap Cp must be multiples of 64; checked by the caller below)
    case 8: return launch_bf<128, 128, 2, 4, NT, 64>(p, vec4, st);
    case 9: return launch_bf<128, 64, 4, 2, NT, 64>(p, vec4, st);
    default: ff_set_error("ff_conv2d_bf16s: bad tile_hint %d", cfg); return FF_ERR_ARG;
  }
}

extern "C" int ff_conv2d_bf16s(const float* in, const void* w_hi, const void* w_lo, int Kp, int Cp, const float* bias,
                               const float* mul, const float* res, float* out, int B, int H, int W, int Cin, int ldi,
                               int Ho, int Wo, int Cout, int ldo, int ldr, int KH, int KW, int sy, int sx, int py, int px,
                               int act, float alpha, int shuffle, int nterms, int tile_hint, const float* kmul, void* stream) {
  FF_CHECK_ARG(in && w_hi && out, "ff_conv2d_bf16s: null pointer");
  FF_CHECK_ARG(!kmul || (KH == 1 && KW == 1 && Cin % 4 == 0 && (((uintptr_t)kmul) & 15) == 0), "ff_conv2d_bf16s: kmul needs a 1x1 kernel, Cin %% 4 == 0 and a 16-byte aligned vector");
  FF_CHECK_ARG(nterms >= 1 && nterms <= 3 && (nterms < 3 || w_lo), "ff_conv2d_bf16s: nterms must be 1..3 (3 needs w_lo)");
  FF_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Ho > 0 && Wo > 0, "ff_conv2d_bf16s: bad dims");
  FF_CHECK_ARG(ldi >= Cin, "ff_conv2d_bf16s: ldi %d < Cin %d", ldi, Cin);
  FF_CHECK_ARG(KH > 0 && KW > 0 && sy > 0 && sx > 0 && py >= 0 && px >= 0, "ff_conv2d_bf16s: bad kernel geometry");
  FF_CHECK_ARG(Kp % 32 == 0 && Kp >= KH * KW * Cin, "ff_conv2d_bf16s: Kp must be K rounded up to 32");
  FF_CHECK_ARG(Cp == 0 || (Cp % 32 == 0 && Cp >= Cin && Kp == KH * KW * Cp), "ff_conv2d_bf16s: TAP layout needs Kp == taps * Cp, Cp = ceil32(Cin)");
  FF_CHECK_ARG((((uintptr_t)w_hi) & 15) == 0 && (!w_lo || (((uintptr_t)w_lo) & 15) == 0), "ff_conv2d_bf16s: weight planes must be 16-byte aligned");
  FF_CHECK_ARG(shuffle == 0 || shuffle == 1 || shuffle == 2, "ff_conv2d_bf16s: shuffle must be 0, 1 (SimpleGate pair product) or 2 (PixelShuffle)");
  FF_CHECK_ARG(shuffle != 2 || Cout % 4 == 0, "ff_conv2d_bf16s: shuffle needs Cout %% 4 == 0");
  FF_CHECK_ARG(shuffle != 1 || (Cout % 8 == 0 && !res && !mul && act == 0 && alpha == 1.f && ldo % 4 == 0 && (((uintptr_t)out) & 15) == 0),
               "ff_conv2d_bf16s: the pair-product epilogue needs Cout %% 8 == 0, no residual / scale / activation and 16-byte aligned output rows");
  FF_CHECK_ARG(ldo >= (shuffle == 2 ? Cout / 4 : shuffle == 1 ? Cout / 2 : Cout), "ff_conv2d_bf16s: ldo too small");
  FF_CHECK_ARG(!res || ldr >= (shuffle ? Cout / 4 : Cout), "ff_conv2d_bf16s: ldr too small");
  FF_CHECK_ARG((long long)B * Ho * Wo < (1LL << 31), "ff_conv2d_bf16s: M overflows int");
  ConvBfParams p;
  p.in = in; p.w_hi = (const __bf16*)w_hi; p.w_lo = (const __bf16*)w_lo; p.bias = bias; p.mul = mul; p.res = res; p.out = out;
  p.kmul = kmul; p.gate = shuffle == 1 ? 1 : 0;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.ldi = ldi;
  p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.ldo = ldo; p.ldr = ldr;
  p.KH = KH; p.KW = KW; p.sy = sy; p.sx = sx; p.py = py; p.px = px;
  p.K = KH * KW * Cin; p.Kp = Kp; p.Cp = Cp; p.M = B * Ho * Wo;
  p.act = act; p.alpha = alpha; p.shuffle = shuffle;
  p.vec_out = shuffle == 0 && Cout % 4 == 0 && ldo % 4 == 0 && (((uintptr_t)out) & 15) == 0 &&
              (!res || (ldr % 4 == 0 && (((uintptr_t)res) & 15) == 0));
  const bool vec4 = (Cin % 4 == 0) && (ldi % 4 == 0) && (((uintptr_t)in & 15) == 0);
  FF_CHECK_ARG(Cp == 0 || vec4, "ff_conv2d_bf16s: TAP layout needs Cin %% 4 == 0 and 16-byte aligned rows");
  FF_CHECK_ARG(!kmul || vec4, "ff_conv2d_bf16s: kmul needs 16-byte aligned input rows");
  hipStream_t st = (hipStream_t)stream;
  int cfg = tile_hint;
  if (cfg <= 0) {
    // measured on MI355X (profiles/r01_gemm_shapes_bf16x3_v2.txt): 128x128 tiles win whenever N pads well or K is long
    const int r192 = Cout % 192, r128 = Cout % 128;
    // r2 (profiles/r02_gemm_shapes_8wave.txt): the 8-wave forms of the 128-row tiles (two waves per SIMD: one wave's operand
    // reads and global-load waits hide behind the other's MFMAs) beat the 4-wave forms on every 1x1 shape: cfg 5 replaces 1, 7 replaces 2
    if (Cout <= 32) cfg = 3;
    else if (Cout <= 64) cfg = 7;
    else if (r128 == 0) cfg = 5;
    // measured (profiles/r01_gemm_shapes_bf16x3_v3.txt): 256x192 wins for N <= 192 or long K; 128x128 for wide N at K = 180
    else if (vec4 && p.M >= 256 * 64 && (r192 == 0 || r192 > 128) && (Cout <= 192 || p.K >= 320)) cfg = 4;
    else if (Cout > 256 || (Cout > 128 && p.K >= 512)) cfg = 5;
    else cfg = 7;
  }
  const bool k64 = Kp % 64 == 0 && (Cp == 0 || Cp % 64 == 0) && vec4;
  if (cfg >= 8 && !k64) cfg = cfg == 8 ? 5 : 7;
  switch (nterms) {
    case 1: return dispatch_bf<1>(p, vec4, cfg, st);
    case 2: return dispatch_bf<2>(p, vec4, cfg, st);
    default: return dispatch_bf<3>(p, vec4, cfg, st);
  }
}

// fp32 [N][K] -> bf16 planes hi/lo [N][Kp], zero filled; lo may be NULL.
// Cp == 0: flat, Kp = ceil32(K).  Cp > 0 (TAP layout): K = taps*Cin, column tap*Cp + ci holds w[n][tap*Cin + ci].
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ w, int N, int K, int Kp, int Cin, int Cp,
                                                         __bf16* __restrict__ hi, __bf16* __restrict__ lo) {
  const long long total = (long long)N * Kp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int kp = (int)(i % Kp);
    const long long n = i / Kp;
    int k = kp;
    bool ok = kp < K;
    if (Cp > 0) { const int tap = kp / Cp, ci = kp - tap * Cp; ok = ci < Cin; k = tap * Cin + ci; }
    const float f = ok ? w[n * K + k] : 0.f;
    const __bf16 h = (__bf16)f;
    hi[i] = h;
    if (lo) lo[i] = (__bf16)(f - (float)h);
  }
}

extern "C" int ff_split_bf16(const float* w, int N, int K, int Kp, int Cin, int Cp, void* hi, void* lo, void* stream) {
  FF_CHECK_ARG(w && hi && N > 0 && K > 0 && Kp % 32 == 0 && Kp >= K, "ff_split_bf16: bad args");
  FF_CHECK_ARG(Cp == 0 || (Cin > 0 && K % Cin == 0 && Cp % 32 == 0 && Cp >= Cin && Kp == (K / Cin) * Cp), "ff_split_bf16: bad TAP layout");
  long long nb = ((long long)N * Kp + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, N, K, Kp, Cin, Cp, (__bf16*)hi, (__bf16*)lo);
  FF_LAUNCH_CHECK("ff_split_bf16");
  return FF_OK;
}

// DAT SGFN tail in one launch (plain bf16):   out = res + fc2( x1 * (dwconv3x3(LayerNorm(x2)) + b_dw) ) + b2
//   dat_arch.py:117-123 (SpatialGate: x1, x2 = chunk(fc1 output); x2 -> LayerNorm -> depth-wise 3x3 -> times x1),
//   :163-170 (fc2), :736 (residual).
// The two-launch form (ff_dwconv3x3_ln -> ff_conv2d 1x1) writes the 94 MB gate product and reads it back, and its strip walker
// fetches the x2 rows 1.5 times: 283 + 189 MB per block, 156 us.  Here the gate product is produced in MFMA operand order and
// consumed at once: x2 (with a one-pixel halo), x1, the residual and the output are the only HBM traffic (~313 MB).
// Dataflow of a 512-thread workgroup = an 8 x 32 token tile (wave = tile row, lane & 31 = column):
//   * the hidden width is walked in 32-channel chunks.  The chunk's x2 halo tile (10 x 34 pixels) is fetched by all threads
//     (128-byte pixel segments), LayerNorm applied on the way (per-pixel mean / rstd from fc1's epilogue, zero outside the image:
//     the convolution pads the NORMALISED tensor) and parked in LDS as fp32 [pixel][32 + 4 pad]; the chunk's nine tap rows and
//     bias follow; fc2's [192][32] bf16 weight tile arrives by LDS-DMA.  All of it is double buffered: chunk c+1 is in flight
//     while chunk c is computed, one workgroup barrier per chunk;
//   * per k-step of 16 channels a lane forms the gate product of ITS token for the 8 channels the MFMA B operand wants
//     (9 taps x 8 channels from the LDS tile, times x1 read straight from global one chunk ahead), converts it to bf16 and
//     feeds six v_mfma_f32_32x32x16_bf16 (out^T[n][token] += W2[n][k] . G[token][k]);
//   * epilogue: + b2 + residual, transposed through LDS into 128-byte row segments.
#include "ff_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define SG_TH 8
#define SG_TW 32
#define SG_HW (SG_TW + 2)                 // halo width 34
#define SG_HP ((SG_TH + 2) * SG_HW)       // 340 halo pixels
#define SG_XROW 36                        // floats per halo pixel in LDS: 32 + 4 pad (144 B: conflict-free ds_read_b128 across pixels)
#define SG_XBUF (SG_HP * SG_XROW * 4)     // 48 960 B
#define SG_WROWB 80                       // fc2 tile row: 32 bf16 + 16 B pad
#define SG_WBUF (15 * 1024)               // 192 rows x 80 B = 15 360 B = 15 DMA pieces
#define SG_DWF (10 * 32)                  // 9 tap rows + bias, 32 channels
#define SG_OFF_X 0
#define SG_OFF_W (2 * SG_XBUF)
#define SG_OFF_DW (SG_OFF_W + 2 * SG_WBUF)
#define SG_OFF_GB (SG_OFF_DW + 2 * SG_DWF * 4)
#define SG_KMAX 512                       // padded hidden width the gamma / beta image holds
#define SG_OFF_ST (SG_OFF_GB + 2 * SG_KMAX * 4)
#define SG_OFF_TOK (SG_OFF_ST + SG_HP * 8)
#define SG_LDS (SG_OFF_TOK + SG_HP * 4)

struct SgfnParams {
  const float* h; const float* stats; const float* gamma; const float* beta; const float* dww; const float* dwb;
  const __bf16* w2; const float* b2; const float* res; float* out;
  int ldh, c2, ldr, ldo, B, H, W, N, HT, ntx, nty;
#ifdef SG_TIMING
  unsigned long long* dbg;   // tools/sg_time.cpp: [block][wave][8] wall-clock stamps (debug build only)
#endif
};
#ifdef SG_TIMING
static unsigned long long* g_sg_dbg = nullptr;
#define SG_T(i) do { if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[((long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define SG_T(i) do { } while (0)
#endif

__global__ __launch_bounds__(512) void sgfn_tail_kernel(SgfnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* GB = reinterpret_cast<float*>(smem + SG_OFF_GB);          // gamma [SG_KMAX] | beta [SG_KMAX], zero padded
  float* ST = reinterpret_cast<float*>(smem + SG_OFF_ST);          // [340][2] mean, rstd
  int* TOK = reinterpret_cast<int*>(smem + SG_OFF_TOK);            // [340] token index or -1
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  int bid = blockIdx.x;
  const int tx = bid % p.ntx; bid /= p.ntx;
  const int ty = bid % p.nty;
  const int b = bid / p.nty;
  const int y0 = ty * SG_TH, x0 = tx * SG_TW;
  SG_T(0);

  // ---- per-workgroup tables: halo pixel -> token (or -1), its LayerNorm statistics; gamma / beta ------------------------------
  if (tid < SG_HP) {
    const int hy = tid / SG_HW, hx = tid - hy * SG_HW;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    const int tk = ok ? (b * p.H + y) * p.W + x : -1;
    TOK[tid] = tk;
    const float2 ms = ok ? *reinterpret_cast<const float2*>(p.stats + 2 * (long long)tk) : (float2){0.f, 0.f};
    ST[2 * tid] = ms.x; ST[2 * tid + 1] = ms.y;
  }
  for (int i = tid; i < SG_KMAX; i += 512) { GB[i] = i < p.c2 ? p.gamma[i] : 0.f; GB[SG_KMAX + i] = i < p.c2 ? p.beta[i] : 0.f; }

  // fc2 tile DMA: piece wid + 8 i of the 15 one-KiB pieces (192 rows x 5 slots; the pad slot re-reads slot 3)
  int woff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int s = (wid + 8 * i) * 64 + lane;
    if (s > 959) s = 959;
    const int row = s / 5;
    int q = s - row * 5;
    if (q > 3) q = 3;
    woff[i] = row * 32 + q * 8;
  }
  auto dma_w = [&](int c, int buf) {
    const __bf16* rec = p.w2 + (long long)c * (192 * 32);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (wid + 8 * i < 15)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + woff[i]),
                                         (__attribute__((address_space(3))) void*)(smem + SG_OFF_W + buf * SG_WBUF + (wid + 8 * i) * 1024), 16, 0, 0);
  };
  dma_w(0, 0);
  __syncthreads();

  // staging role: channel quad q4 of halo pixels ps + 64 i; compute role: token (row wid, column l31)
  const int sq = tid & 7, ps = tid >> 3;
  int stok[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) stok[i] = (ps + 64 * i < SG_HP) ? TOK[ps + 64 * i] : -1;
  const int cy = y0 + wid, cx = x0 + l31;
  const bool cvalid = cy < p.H && cx < p.W;
  const long long ctok = cvalid ? (long long)(b * p.H + cy) * p.W + cx : 0;
  const float* x1row = p.h + ctok * p.ldh + 8 * hh;

  f32x4 stg[6];
  float dwv = 0.f;
  auto stage_load = [&](int c) {
    const int ch = c * 32 + 4 * sq;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const bool ok = stok[i] >= 0 && ch < p.c2;
      stg[i] = *reinterpret_cast<const f32x4*>(p.h + (ok ? (long long)stok[i] * p.ldh + p.c2 + ch : 0));
    }
    if (tid < SG_DWF) {
      const int r = tid >> 5, cc = c * 32 + (tid & 31);
      dwv = cc < p.c2 ? (r < 9 ? p.dww[(long long)r * p.c2 + cc] : (p.dwb ? p.dwb[cc] : 0.f)) : 0.f;
    }
  };
  auto stage_store = [&](int c, int buf) {
    const int ch = c * 32 + 4 * sq;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(GB + ch), b4 = *reinterpret_cast<const f32x4*>(GB + SG_KMAX + ch);
    float* xs = reinterpret_cast<float*>(smem + SG_OFF_X + buf * SG_XBUF);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int hp = ps + 64 * i;
      if (hp < SG_HP) {
        const float mean = ST[2 * hp], rstd = ST[2 * hp + 1];
        f32x4 t = (stg[i] - mean) * rstd * g4 + b4;
        if (stok[i] < 0 || ch >= p.c2) t = (f32x4){0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(xs + hp * SG_XROW + 4 * sq) = t;
      }
    }
    if (tid < SG_DWF) reinterpret_cast<float*>(smem + SG_OFF_DW)[buf * SG_DWF + tid] = dwv;
  };
  f32x4 x1c[4], x1n[4];
  auto x1_load = [&](int c, f32x4 (&d)[4]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      d[2 * s] = *reinterpret_cast<const f32x4*>(x1row + c * 32 + 16 * s);
      d[2 * s + 1] = *reinterpret_cast<const f32x4*>(x1row + c * 32 + 16 * s + 4);
    }
  };

  SG_T(1);
  stage_load(0);
  x1_load(0, x1c);
  stage_store(0, 0);
  f32x16 oacc[6];
#pragma unroll
  for (int n = 0; n < 6; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[n][r] = 0.f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  SG_T(2);
  for (int c = 0; c < p.HT; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < p.HT;
    if (more) {
      dma_w(c + 1, buf ^ 1);
      stage_load(c + 1);
      x1_load(c + 1, x1n);
    }
    const float* xs = reinterpret_cast<const float*>(smem + SG_OFF_X + buf * SG_XBUF) + (wid * SG_HW + l31) * SG_XROW + 8 * hh;
    const float* dw = reinterpret_cast<const float*>(smem + SG_OFF_DW) + buf * SG_DWF + 8 * hh;
    const unsigned char* wt = smem + SG_OFF_W + buf * SG_WBUF + l31 * SG_WROWB + 16 * hh;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 ga = *reinterpret_cast<const f32x4*>(dw + 9 * 32 + 16 * s), gb = *reinterpret_cast<const f32x4*>(dw + 9 * 32 + 16 * s + 4);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* xp = xs + (ky * SG_HW + kx) * SG_XROW + 16 * s;
          const float* wp = dw + (ky * 3 + kx) * 32 + 16 * s;
          ga += *reinterpret_cast<const f32x4*>(xp) * *reinterpret_cast<const f32x4*>(wp);
          gb += *reinterpret_cast<const f32x4*>(xp + 4) * *reinterpret_cast<const f32x4*>(wp + 4);
        }
      ga *= x1c[2 * s];
      gb *= x1c[2 * s + 1];
      bf16x8 g;
#pragma unroll
      for (int e = 0; e < 4; ++e) { g[e] = (__bf16)ga[e]; g[4 + e] = (__bf16)gb[e]; }
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(wt + n * 32 * SG_WROWB + 32 * s);
        oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, g, oacc[n], 0, 0, 0);
      }
    }
    if (more) {
      stage_store(c + 1, buf ^ 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) x1c[i] = x1n[i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // chunk c+1 is staged and its fc2 tile has landed; every wave is done with chunk c
  }

  SG_T(3);
  // ---- epilogue: + b2 + residual, through the wave's LDS patch (the halo buffers are idle) into 128-byte row segments ---------
  {
    float* patch = reinterpret_cast<float*>(smem) + wid * (32 * 36);
    const int tq = lane >> 3, q4 = 4 * (lane & 7);
    const bool rowok = cy < p.H;
    const long long tok0 = (long long)(b * p.H + (rowok ? cy : 0)) * p.W + x0;
    // every residual quad of the wave's 32 x 192 output is requested up front (tools/sg_time.cpp: fetched tile by tile after the
    // transposition each one exposed its HBM latency -- the epilogue was 26.5 us of the launch's 93)
    f32x4 rq[6][4];
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = tq + 8 * i, c4 = n * 32 + q4;
        const bool ok = p.res && c4 < p.N && rowok && x0 + t < p.W;
        rq[n][i] = ok ? *reinterpret_cast<const f32x4*>(p.res + (tok0 + t) * p.ldr + c4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const int c4 = n * 32 + q4;
      const bool cok = c4 < p.N;
      const f32x4 b4 = p.b2 ? *reinterpret_cast<const f32x4*>(p.b2 + (cok ? c4 : 0)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = oacc[n][4 * g + e];
        *reinterpret_cast<f32x4*>(patch + l31 * 36 + 8 * g + 4 * hh) = v4;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = tq + 8 * i;
        const f32x4 ov = *reinterpret_cast<const f32x4*>(patch + t * 36 + q4) + b4 + rq[n][i];
        if (cok && rowok && x0 + t < p.W) *reinterpret_cast<f32x4*>(p.out + (tok0 + t) * p.ldo + c4) = ov;
      }
    }
  }
  SG_T(4);
#ifdef SG_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SG_T(5);
#endif
}

extern "C" int ff_sgfn_tail(const float* h, int ldh, int c2, const float* stats, const float* gamma, const float* beta,
                            const float* dw_tapmajor, const float* dw_bias, const void* fc2_tiles, int hidden_tiles, const float* b2,
                            const float* res, int ldr, float* out, int ldo, int B, int H, int W, int N, void* stream) {
  FF_CHECK_ARG(h && stats && gamma && beta && dw_tapmajor && fc2_tiles && out, "ff_sgfn_tail: null pointer");
  FF_CHECK_ARG(B > 0 && H > 0 && W > 0 && c2 > 0 && c2 % 4 == 0 && ldh >= 2 * c2 && ldh % 4 == 0 && N > 0 && N <= 192 && N % 4 == 0,
               "ff_sgfn_tail: needs hidden % 4 == 0, rows of >= 2 * hidden floats, N <= 192 (N %% 4 == 0)");
  FF_CHECK_ARG(hidden_tiles == (c2 + 31) / 32 && hidden_tiles * 32 <= SG_KMAX && hidden_tiles * 32 <= ldh, "ff_sgfn_tail: hidden_tiles must be ceil(hidden / 32) <= 16 (and 32 * hidden_tiles <= ldh)");
  FF_CHECK_ARG(ldo >= N && ldo % 4 == 0 && (!res || (ldr >= N && ldr % 4 == 0)), "ff_sgfn_tail: out / res rows must be 16-byte aligned");
  FF_CHECK_ARG((((uintptr_t)h) & 15) == 0 && (((uintptr_t)out) & 15) == 0 && (((uintptr_t)fc2_tiles) & 15) == 0 && (((uintptr_t)stats) & 7) == 0 &&
               (!res || (((uintptr_t)res) & 15) == 0) && (!b2 || (((uintptr_t)b2) & 15) == 0), "ff_sgfn_tail: alignment");
  FF_CHECK_ARG((long long)B * H * W * ldh < (1LL << 31) && out != h, "ff_sgfn_tail: tensor too large for 32-bit token offsets / in-place");
  SgfnParams p;
  p.h = h; p.stats = stats; p.gamma = gamma; p.beta = beta; p.dww = dw_tapmajor; p.dwb = dw_bias; p.w2 = (const __bf16*)fc2_tiles;
  p.b2 = b2; p.res = res; p.out = out; p.ldh = ldh; p.c2 = c2; p.ldr = ldr; p.ldo = ldo; p.B = B; p.H = H; p.W = W; p.N = N; p.HT = hidden_tiles;
  p.ntx = (W + SG_TW - 1) / SG_TW; p.nty = (H + SG_TH - 1) / SG_TH;
#ifdef SG_TIMING
  p.dbg = g_sg_dbg;
#endif
  const long long nblk = (long long)B * p.ntx * p.nty;
  FF_CHECK_ARG(nblk < (1LL << 31), "ff_sgfn_tail: grid too large");
  static_assert(SG_LDS <= 160 * 1024, "LDS image too large");
  static_assert(8 * 32 * 36 * 4 <= 2 * SG_XBUF, "epilogue patches must fit in the halo buffers");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sgfn_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SG_LDS);
    if (e != hipSuccess) { ff_set_error("ff_sgfn_tail: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; }
    attr_set = true;
  }
  hipLaunchKernelGGL(sgfn_tail_kernel, dim3((unsigned)nblk), dim3(512), SG_LDS, (hipStream_t)stream, p);
  FF_LAUNCH_CHECK("ff_sgfn_tail");
  return FF_OK;
}

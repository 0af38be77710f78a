// Fused transformer MLP on tokens:  out = x + fc2( GELU( fc1( LayerNorm(x) ) ) )      (hat_arch.py:83-94,307;
// the HAB / OCAB feed-forward, 84 per HAT forward).  One launch instead of LayerNorm + 2 GEMMs, and the
// hidden activation (360 floats per token) never leaves the CU: HBM traffic is x in, out out (94 MB per call at
// 65 536 tokens instead of 423 MB), which is what bounds these K=180 layers.
//
// "Flash-MLP" dataflow, split-operand bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate, bf16x3):
//   * a wave owns 32 tokens; LayerNorm(x) of those tokens is the B operand of every k-step (lane = token, 8 consecutive
//     k per lane and step), produced once: the hi halves live in REGISTERS (48 VGPRs), the lo halves in a wave-private
//     LDS image (keeping both in registers next to the 96 output accumulators overflows the 256-VGPR budget);
//   * per 32-wide hidden tile:  H^T[hidden][token] = W1 . LN(x)^T  (A = W1 rows from LDS), bias, GELU, split to bf16;
//     the accumulator has hidden on the registers and the token on the lane, so it IS the B operand of
//     OUT^T[n][token] += W2[n][hidden] . H^T   (guide section 3, 'An accumulator tile as the next MFMA's operand';
//     W2's hidden columns are stored in the matching permuted order: bits 2 and 3 of the index swapped);
//   * weight tiles (W1: 32 x 192, W2: 192 x 32, hi + lo planes) are shared by the 8 waves of the workgroup and filled by
//     LDS-DMA (global_load_lds_dwordx4: no VGPRs, no staging stores).  The LDS images are 25 + 30 lane-linear 1-KiB
//     pieces; rows are padded by one 16-byte slot (conflict-free ds_read_b128) and the per-lane SOURCE address does the
//     row/pad bookkeeping (guide section 5, Caveat).  Two barriers per hidden tile: W1 is refilled for tile t+1 while
//     GELU + GEMM2 of tile t run, W2 while GEMM1 of tile t+1 runs;
//   * epilogue: 32x32 tiles are transposed through LDS so global stores / residual loads are 128-byte row segments.
// Workgroup = 8 waves = 256 tokens: 65 536 tokens -> 256 workgroups = one per CU.
#include "ff_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define TM_KP 192      // padded K (model dim 180)
#define TM_NP 192      // padded N
#define TM_KS 12       // k-steps of 16
#define W1SLOTS 25     // 16-byte slots per W1 row: 24 data (192 bf16) + 1 pad  -> 400 B
#define W2SLOTS 5      // slots per W2 row: 4 data (32 bf16) + 1 pad            -> 80 B
#define W1ROWB (W1SLOTS * 16)
#define W2ROWB (W2SLOTS * 16)
#define W1PL (32 * W1SLOTS)          // slots per W1 plane  (800)
#define W2PL (TM_NP * W2SLOTS)       // slots per W2 plane  (960)
#define W1PIECES (2 * W1PL / 64)      // 25
#define W2PIECES (2 * W2PL / 64)      // 30
#define TILE_ELEMS 6144              // bf16 elements per plane per hidden tile (32*192 == 192*32)
#define XLROWB 400                   // x_lo image row (per token): 192 bf16 + pad

struct TokenMlpParams {
  const float* x; float* out;
  const float* gamma; const float* beta;
  const __bf16* w;        // [HT][4 planes: W1hi, W1lo, W2hi, W2lo][6144]; W2 planes are [192][32] with permuted hidden columns
  const float* b1;        // [HT*32] zero padded
  const float* b2;        // [N]
  long long M;
  int ldx, ldo, K, N, HT;
  float eps;
};

template <bool VEC4, int NTERMS>
__global__ __launch_bounds__(512) void token_mlp_kernel(TokenMlpParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W1B = W1PL * 16, W2B = W2PL * 16;
  unsigned char* W1s = smem;                         // [2 planes][32][400]
  unsigned char* W2s = smem + 2 * W1B;               // [2 planes][192][80]
  unsigned char* XLs = smem + 2 * W1B + 2 * W2B;     // [8 waves][32 tokens][400]
  float* B1s = reinterpret_cast<float*>(XLs + 8 * 32 * XLROWB);   // [HT*32] fc1 bias (read every hidden tile)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const long long tok = (long long)blockIdx.x * 256 + wid * 32 + l31;
  const bool tvalid = tok < p.M;

  // ---- LDS-DMA bookkeeping: this wave fills pieces wid, wid+8, ... of each image; per lane the source offset in a tile record
  int off1[4], off2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int s = (wid + 8 * i) * 64 + lane;               // slot in the W1 image (2 planes x 32 rows x 25 slots)
    if (s >= 2 * W1PL) s = 2 * W1PL - 1;
    int plane = s / W1PL, t = s - plane * W1PL, row = t / W1SLOTS, q = t - row * W1SLOTS;
    if (q > 23) q = 23;
    off1[i] = plane * TILE_ELEMS + row * 192 + q * 8;
    s = (wid + 8 * i) * 64 + lane;                   // slot in the W2 image (2 planes x 192 rows x 5 slots)
    if (s >= 2 * W2PL) s = 2 * W2PL - 1;
    plane = s / W2PL; t = s - plane * W2PL; row = t / W2SLOTS; q = t - row * W2SLOTS;
    if (q > 3) q = 3;
    off2[i] = (2 + plane) * TILE_ELEMS + row * 32 + q * 8;
  }
  auto dma_w1 = [&](int ht) {
    const __bf16* rec = p.w + (long long)ht * (4 * TILE_ELEMS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (wid + 8 * i < W1PIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + off1[i]),
                                         (__attribute__((address_space(3))) void*)(W1s + (wid + 8 * i) * 1024), 16, 0, 0);
  };
  auto dma_w2 = [&](int ht) {
    const __bf16* rec = p.w + (long long)ht * (4 * TILE_ELEMS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (wid + 8 * i < W2PIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + off2[i]),
                                         (__attribute__((address_space(3))) void*)(W2s + (wid + 8 * i) * 1024), 16, 0, 0);
  };
  dma_w1(0);
  dma_w2(0);
  for (int i = tid; i < p.HT * 32; i += 512) B1s[i] = p.b1[i];

  // ---- LayerNorm(x) of this lane's token half -> bf16 hi (registers) / lo (wave-private LDS rows) -----------------
  bf16x8 xh[TM_KS];
  unsigned char* xl_row = XLs + (size_t)(wid * 32 + l31) * XLROWB + 16 * hh;     // + 32*st per k-step
  {
    float v[TM_KS][8];
    float s = 0.f;
    // staging patch = this wave's own x_lo rows (12.8 KB, written only below, after the fragments are in registers)
    ff_wave_rows_to_frags<3>(p.x, p.ldx, (long long)blockIdx.x * 256 + wid * 32, p.M, p.K,
                          reinterpret_cast<float*>(XLs + (size_t)wid * 32 * XLROWB), lane, v);
#pragma unroll
    for (int st = 0; st < TM_KS; ++st)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[st][j];
    s += __shfl_xor(s, 32);
    const float mean = s / (float)p.K;
    float qv = 0.f;
#pragma unroll
    for (int st = 0; st < TM_KS; ++st)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * st + 8 * hh + j;
        const float d = (k < p.K) ? v[st][j] - mean : 0.f;
        qv += d * d;
      }
    qv += __shfl_xor(qv, 32);
    const float rstd = 1.0f / sqrtf(qv / (float)p.K + p.eps);
#pragma unroll
    for (int st = 0; st < TM_KS; ++st) {
      const int k0 = 16 * st + 8 * hh;
      const int ka = k0 < p.K ? k0 : 0, kb = k0 + 4 < p.K ? k0 + 4 : 0;     // clamped: values beyond K are unused
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + ka), b0 = *reinterpret_cast<const f32x4*>(p.beta + ka);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(p.gamma + kb), b1v = *reinterpret_cast<const f32x4*>(p.beta + kb);
      bf16x8 lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const float gg = j < 4 ? g0[j & 3] : g1[j & 3], bb = j < 4 ? b0[j & 3] : b1v[j & 3];
        const float f = (k < p.K) ? (v[st][j] - mean) * rstd * gg + bb : 0.f;
        const __bf16 h = (__bf16)f;
        xh[st][j] = h;
        lo[j] = (__bf16)(f - (float)h);
      }
      *reinterpret_cast<bf16x8*>(xl_row + 32 * st) = lo;
    }
  }

  f32x16 oacc[6];
#pragma unroll
  for (int n = 0; n < 6; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[n][r] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile 0 landed (this wave's pieces)
  __syncthreads();

  for (int ht = 0; ht < p.HT; ++ht) {
    // ---- H^T tile = b1 + W1 . LN(x)^T ------------------------------------------------------------------------
    f32x16 hacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = B1s[ht * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
    {
      // operands are read TWO k-steps ahead (hipcc otherwise issues ds_read; s_waitcnt lgkmcnt(0); mfma per step and
      // exposes the LDS latency 24 times per tile); sched_barrier pins the read block in front of the MFMA block
      const unsigned char* ap = W1s + l31 * W1ROWB + 16 * hh;
      bf16x8 fa[2], fl[2], fx[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        fa[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u);
        fl[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u + W1B);
        fx[u] = *reinterpret_cast<const bf16x8*>(xl_row + 32 * u);
      }
#pragma unroll
      for (int st = 0; st < TM_KS; ++st) {
        const bf16x8 ah = fa[st & 1], al = fl[st & 1], xl = fx[st & 1];
        if (st + 2 < TM_KS) {
          fa[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2));
          fl[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2) + W1B);
          fx[st & 1] = *reinterpret_cast<const bf16x8*>(xl_row + 32 * (st + 2));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NTERMS == 3) {
          hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, hacc, 0, 0, 0);
          hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[st], hacc, 0, 0, 0);
        }
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[st], hacc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // W2(ht) pieces of this wave have landed
    __syncthreads();                                    // A: every wave finished reading W1; W2(ht) visible to all
    if (ht + 1 < p.HT) dma_w1(ht + 1);
    // ---- GELU, split; registers 8s..8s+7 are the B fragment of k-step s ----------------------------------------
    bf16x8 gh[2], gl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = ff_gelu_fast(hacc[8 * s + j]);
        const __bf16 h = (__bf16)g;
        gh[s][j] = h;
        gl[s][j] = (__bf16)(g - (float)h);
      }
    // ---- OUT^T += W2 . H^T ---------------------------------------------------------------------------------------
    {
      const unsigned char* ap = W2s + l31 * W2ROWB + 16 * hh;        // step u = 2n + s: row n*32 + l31, k-step s
      bf16x8 fa[2], fl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        fa[u] = *reinterpret_cast<const bf16x8*>(ap + (u >> 1) * 32 * W2ROWB + 32 * (u & 1));
        fl[u] = *reinterpret_cast<const bf16x8*>(ap + (u >> 1) * 32 * W2ROWB + 32 * (u & 1) + W2B);
      }
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const int n = u >> 1, s2 = u & 1;
        const bf16x8 ah = fa[u & 1], al = fl[u & 1];
        if (u + 2 < 12) {
          fa[u & 1] = *reinterpret_cast<const bf16x8*>(ap + ((u + 2) >> 1) * 32 * W2ROWB + 32 * ((u + 2) & 1));
          fl[u & 1] = *reinterpret_cast<const bf16x8*>(ap + ((u + 2) >> 1) * 32 * W2ROWB + 32 * ((u + 2) & 1) + W2B);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NTERMS == 3) {
          oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gl[s2], oacc[n], 0, 0, 0);
          oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gh[s2], oacc[n], 0, 0, 0);
        }
        oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh[s2], oacc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // W1(ht+1) pieces of this wave have landed
    __syncthreads();                                    // B: every wave finished reading W2; W1(ht+1) visible to all
    if (ht + 1 < p.HT) dma_w2(ht + 1);
  }

  // ---- epilogue: all 96 residual loads first (x fragments are dead, registers are free), then per n-tile a
  // transpose through LDS -> coalesced 128-byte row segments.  (out may alias x as far as the compiler knows: a
  // load placed after a store would be serialised behind it.)
  const long long tok0 = (long long)blockIdx.x * 256 + wid * 32;
  if (VEC4) {
    // 16-byte form: the accumulator's four consecutive channels per r >> 2 go to the patch as one ds_write_b128, a lane
    // then owns (token row lane >> 3 (+8 per step), channel quad lane & 7): one float4 residual load and one float4
    // store per step = 8 whole 128-byte row segments per instruction (2 with dword accesses)
    float* tr = reinterpret_cast<float*>(smem) + wid * (32 * 36);    // per-wave [token][36] inside the (now idle) W images
    const int tq = lane >> 3, q4 = 4 * (lane & 7);
    f32x4 rq[6][4];
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const int c4 = n * 32 + q4;
      const bool cok = c4 < p.N;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.b2 + (cok ? c4 : 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long tk = tok0 + tq + 8 * i;
        const f32x4 u = *reinterpret_cast<const f32x4*>(p.x + (tk < p.M ? tk : 0) * p.ldx + (cok ? c4 : 0));
        rq[n][i] = u + b4;
      }
    }
#pragma unroll
    for (int n = 0; n < 6; ++n) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = oacc[n][4 * g + e];
        *reinterpret_cast<f32x4*>(tr + l31 * 36 + 8 * g + 4 * hh) = v4;
      }
      f32x4 ov[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ov[i] = *reinterpret_cast<const f32x4*>(tr + (tq + 8 * i) * 36 + q4) + rq[n][i];
      const int c4 = n * 32 + q4;
      if (c4 < p.N) {
        float* op = p.out + (tok0 + tq) * p.ldo + c4;
        if (tok0 + 32 <= p.M) {
#pragma unroll
          for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(op + (long long)(8 * i) * p.ldo) = ov[i];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (tok0 + tq + 8 * i < p.M) *reinterpret_cast<f32x4*>(op + (long long)(8 * i) * p.ldo) = ov[i];
        }
      }
    }
    return;
  }
  float* tr = reinterpret_cast<float*>(smem) + wid * (32 * 33);      // per-wave [token][33] inside the (now idle) W images
  float rv[6][16];
#pragma unroll
  for (int n = 0; n < 6; ++n) {
    const int col = n * 32 + l31;
    const int cc = col < p.N ? col : 0;
    const float bias = col < p.N ? p.b2[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long long tk = tok0 + 2 * i + hh;
      rv[n][i] = p.x[(tk < p.M ? tk : 0) * p.ldx + cc] + bias;
    }
  }
#pragma unroll
  for (int n = 0; n < 6; ++n) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tr[l31 * 33 + (r & 3) + 8 * (r >> 2) + 4 * hh] = oacc[n][r];
    const int col = n * 32 + l31;
    float ov[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ov[i] = tr[(2 * i + hh) * 33 + l31] + rv[n][i];
    if (col < p.N) {
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
}

// ----------------------------------------------------------------------------------------------------------------------
// Attention output projection + residuals + norm2 + MLP in ONE launch (HAB / OCAB tail, hat_arch.py:303-307):
//     x1  = shortcut + proj(att) + conv_x * conv_scale            (:305; the conv term is absent in OCAB :436)
//     out = x1 + fc2(GELU(fc1(LayerNorm(x1))))                    (:307 / :437)
// The two-kernel form (token_linear proj -> token_mlp) wrote x1 (47 MB) and read it back; here the six 32-channel tiles of
// proj^T[channel][token] stay in the 96 output accumulators: the residuals are added there, the LayerNorm statistics are taken
// from them (a lane holds 96 of its token's 192 padded channels, the other half sits in lane + 32), the normalised values
// become the fc1 operand straight from the accumulator registers -- registers 8s..8s+7 of tile n are k-step 2n+s, so fc1's
// K columns are stored in the matching permuted order (prep.pack_token_projmlp) -- and x1 itself remains in the
// accumulators as the final residual.  att rows are the B operand of the projection (hi in registers, lo in the wave's
// LDS rows, like x in token_mlp); the projection's weight tiles use the W1 / W2 image areas as a 2-slot ring.
struct TokenProjMlpParams {
  const float* att; const float* x; const float* c2; const float* rs2;   // c2 / rs2 may be NULL (OCAB)
  float* out;
  const float* gamma; const float* beta;
  const __bf16* wp;       // proj tiles [6][2 planes][32][192]
  const float* bp;        // [192] proj bias, zero padded
  const __bf16* w;        // MLP tiles as in TokenMlpParams, W1 columns permuted
  const float* b1; const float* b2;
  long long M;
  int lda, ldx, ldc, ldo, K, N, HT;
  float eps;
  int io_bf16;            // plain-bf16 kernel only: bit 0 = att rows are bf16 (lda in elements), bit 1 = c2 rows are bf16 (ldc in elements)
#ifdef TM_TIMING
  unsigned long long* dbg;   // tools/pm_time.cpp: [block][wave][64] wall-clock stamps (debug build only)
#endif
};
#ifdef TM_TIMING
static unsigned long long* g_tm_dbg = nullptr;
#define TM_T(i) do { if (p.dbg && lane == 0) p.dbg[((long long)blockIdx.x * 8 + wid) * 64 + (i)] = wall_clock64(); } while (0)
#else
#define TM_T(i) do { } while (0)
#endif

template <int NTERMS>
__global__ __launch_bounds__(512) void token_projmlp_kernel(TokenProjMlpParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W1B = W1PL * 16, W2B = W2PL * 16;
  unsigned char* W1s = smem;
  unsigned char* W2s = smem + 2 * W1B;
  unsigned char* XLs = smem + 2 * W1B + 2 * W2B;
  float* B1s = reinterpret_cast<float*>(XLs + 8 * 32 * XLROWB);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const long long tok0 = (long long)blockIdx.x * 256 + wid * 32;

  int off1[4], off2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int s = (wid + 8 * i) * 64 + lane;
    if (s >= 2 * W1PL) s = 2 * W1PL - 1;
    int plane = s / W1PL, t = s - plane * W1PL, row = t / W1SLOTS, q = t - row * W1SLOTS;
    if (q > 23) q = 23;
    off1[i] = plane * TILE_ELEMS + row * 192 + q * 8;
    s = (wid + 8 * i) * 64 + lane;
    if (s >= 2 * W2PL) s = 2 * W2PL - 1;
    plane = s / W2PL; t = s - plane * W2PL; row = t / W2SLOTS; q = t - row * W2SLOTS;
    if (q > 3) q = 3;
    off2[i] = (2 + plane) * TILE_ELEMS + row * 32 + q * 8;
  }
  auto dma_w1 = [&](int ht) {
    const __bf16* rec = p.w + (long long)ht * (4 * TILE_ELEMS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (wid + 8 * i < W1PIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + off1[i]),
                                         (__attribute__((address_space(3))) void*)(W1s + (wid + 8 * i) * 1024), 16, 0, 0);
  };
  auto dma_w2 = [&](int ht) {
    const __bf16* rec = p.w + (long long)ht * (4 * TILE_ELEMS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (wid + 8 * i < W2PIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + off2[i]),
                                         (__attribute__((address_space(3))) void*)(W2s + (wid + 8 * i) * 1024), 16, 0, 0);
  };
  // projection tile n (a W1-format image) into ring slot 0 = the W1 area, 1 = the W2 area (30 720 B >= 25 600 B)
  auto dma_proj = [&](int n, int slot) {
    const __bf16* rec = p.wp + (long long)n * (2 * TILE_ELEMS);
    unsigned char* dst = slot ? W2s : W1s;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (wid + 8 * i < W1PIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + off1[i]),
                                         (__attribute__((address_space(3))) void*)(dst + (wid + 8 * i) * 1024), 16, 0, 0);
  };
  TM_T(0);
  dma_proj(0, 0);
  for (int i = tid; i < p.HT * 32; i += 512) B1s[i] = p.b1[i];

  // ---- att rows -> split fragments (hi registers, lo in the wave's LDS rows) ---------------------------------------
  bf16x8 xh[TM_KS];
  unsigned char* xl_row = XLs + (size_t)(wid * 32 + l31) * XLROWB + 16 * hh;
  {
    float v[TM_KS][8];
    ff_wave_rows_to_frags<3>(p.att, p.lda, tok0, p.M, p.K, reinterpret_cast<float*>(XLs + (size_t)wid * 32 * XLROWB), lane, v);
#pragma unroll
    for (int st = 0; st < TM_KS; ++st) {
      bf16x8 lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = v[st][j];
        const __bf16 h = (__bf16)f;
        xh[st][j] = h;
        lo[j] = (__bf16)(f - (float)h);
      }
      *reinterpret_cast<bf16x8*>(xl_row + 32 * st) = lo;
    }
  }

  TM_T(1);
  // ---- proj^T tiles into the output accumulators -------------------------------------------------------------------
  f32x16 oacc[6];
#pragma unroll
  for (int n = 0; n < 6; ++n) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // tile n landed everywhere; every wave is past tile n-1
    if (n + 1 < 6) dma_proj(n + 1, (n + 1) & 1);
    const float* bpn = p.bp + n * 32 + 4 * hh;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[n][r] = bpn[(r & 3) + 8 * (r >> 2)];
    const unsigned char* ap = ((n & 1) ? W2s : W1s) + l31 * W1ROWB + 16 * hh;
    bf16x8 fa[2], fl[2], fx[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      fa[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u);
      fl[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u + W1B);
      fx[u] = *reinterpret_cast<const bf16x8*>(xl_row + 32 * u);
    }
#pragma unroll
    for (int st = 0; st < TM_KS; ++st) {
      const bf16x8 ah = fa[st & 1], al = fl[st & 1], xl = fx[st & 1];
      if (st + 2 < TM_KS) {
        fa[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2));
        fl[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2) + W1B);
        fx[st & 1] = *reinterpret_cast<const bf16x8*>(xl_row + 32 * (st + 2));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (NTERMS == 3) {
        oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, oacc[n], 0, 0, 0);
        oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[st], oacc[n], 0, 0, 0);
      }
      oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[st], oacc[n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  TM_T(2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // both image areas are free again: start the MLP's first tiles
  TM_T(3);
  dma_w1(0);
  dma_w2(0);

  // ---- residuals: coalesced float4 loads (lane = token row lane>>3 (+8i), channel quad) go through the wave's own LDS rows
  //      (the att_lo image is dead) and come back in accumulator order: x1 = proj + shortcut + conv * scale ------------
  {
    float* tr = reinterpret_cast<float*>(XLs + (size_t)wid * 32 * XLROWB);      // [32 tokens][36]
    const int tq = lane >> 3, q4 = 4 * (lane & 7);
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const int c4 = n * 32 + q4;
      const bool cok = c4 < p.N;
      f32x4 sc = {0.f, 0.f, 0.f, 0.f};
      if (p.c2 && cok) sc = *reinterpret_cast<const f32x4*>(p.rs2 + c4);
      f32x4 rq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long tk = tok0 + tq + 8 * i;
        const bool ok = cok && tk < p.M;
        f32x4 r0 = *reinterpret_cast<const f32x4*>(p.x + (ok ? tk * p.ldx + c4 : 0));
        if (p.c2) r0 += *reinterpret_cast<const f32x4*>(p.c2 + (ok ? tk * p.ldc + c4 : 0)) * sc;
        rq[i] = ok ? r0 : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(tr + (tq + 8 * i) * 36 + q4) = rq[i];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(tr + l31 * 36 + 8 * g + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) oacc[n][4 * g + e] += u[e];
      }
    }
  }
  TM_T(4);
  // ---- LayerNorm(x1) from the accumulators -> fc1 operand (hi registers, lo rows) -------------------------------------
  {
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += (n * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh < p.K) ? oacc[n][r] : 0.f;
    s += __shfl_xor(s, 32);
    const float mean = s / (float)p.K;
    float qv = 0.f;
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = (n * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh < p.K) ? oacc[n][r] - mean : 0.f;
        qv += d * d;
      }
    qv += __shfl_xor(qv, 32);
    const float rstd = 1.0f / sqrtf(qv / (float)p.K + p.eps);
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 lo;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const int c0 = n * 32 + 8 * (2 * s2 + g2) + 4 * hh;           // channels of registers 8 s2 + 4 g2 .. + 3
          const bool cok = c0 < p.K;
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gamma + (cok ? c0 : 0));
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.beta + (cok ? c0 : 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float f = cok ? (oacc[n][8 * s2 + 4 * g2 + e] - mean) * rstd * g4[e] + b4[e] : 0.f;
            const __bf16 h = (__bf16)f;
            xh[2 * n + s2][4 * g2 + e] = h;
            lo[4 * g2 + e] = (__bf16)(f - (float)h);
          }
        }
        *reinterpret_cast<bf16x8*>(xl_row + 32 * (2 * n + s2)) = lo;
      }
  }
  TM_T(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  TM_T(6);

  // ---- the MLP over hidden tiles: identical to token_mlp_kernel (W1 columns are permuted to the operand order above) ----
  for (int ht = 0; ht < p.HT; ++ht) {
    f32x16 hacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = B1s[ht * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
    {
      const unsigned char* ap = W1s + l31 * W1ROWB + 16 * hh;
      bf16x8 fa[2], fl[2], fx[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        fa[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u);
        fl[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u + W1B);
        fx[u] = *reinterpret_cast<const bf16x8*>(xl_row + 32 * u);
      }
#pragma unroll
      for (int st = 0; st < TM_KS; ++st) {
        const bf16x8 ah = fa[st & 1], al = fl[st & 1], xl = fx[st & 1];
        if (st + 2 < TM_KS) {
          fa[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2));
          fl[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2) + W1B);
          fx[st & 1] = *reinterpret_cast<const bf16x8*>(xl_row + 32 * (st + 2));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NTERMS == 3) {
          hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, hacc, 0, 0, 0);
          hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[st], hacc, 0, 0, 0);
        }
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[st], hacc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ht < 3) TM_T(8 + 4 * ht);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ht < 3) TM_T(9 + 4 * ht);
    if (ht + 1 < p.HT) dma_w1(ht + 1);
    bf16x8 gh[2], gl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = ff_gelu_fast(hacc[8 * s + j]);
        const __bf16 h = (__bf16)g;
        gh[s][j] = h;
        gl[s][j] = (__bf16)(g - (float)h);
      }
    {
      const unsigned char* ap = W2s + l31 * W2ROWB + 16 * hh;
      bf16x8 fa[2], fl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        fa[u] = *reinterpret_cast<const bf16x8*>(ap + (u >> 1) * 32 * W2ROWB + 32 * (u & 1));
        fl[u] = *reinterpret_cast<const bf16x8*>(ap + (u >> 1) * 32 * W2ROWB + 32 * (u & 1) + W2B);
      }
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const int n = u >> 1, s2 = u & 1;
        const bf16x8 ah = fa[u & 1], al = fl[u & 1];
        if (u + 2 < 12) {
          fa[u & 1] = *reinterpret_cast<const bf16x8*>(ap + ((u + 2) >> 1) * 32 * W2ROWB + 32 * ((u + 2) & 1));
          fl[u & 1] = *reinterpret_cast<const bf16x8*>(ap + ((u + 2) >> 1) * 32 * W2ROWB + 32 * ((u + 2) & 1) + W2B);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NTERMS == 3) {
          oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gl[s2], oacc[n], 0, 0, 0);
          oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gh[s2], oacc[n], 0, 0, 0);
        }
        oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh[s2], oacc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ht < 3) TM_T(10 + 4 * ht);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ht < 3) TM_T(11 + 4 * ht);
    if (ht + 1 < p.HT) dma_w2(ht + 1);
  }
  TM_T(40);

  // ---- epilogue: out = accumulators (x1 + fc2 part) + b2, transposed through LDS into 128-byte row segments ------------
  {
    float* tr = reinterpret_cast<float*>(smem) + wid * (32 * 36);
    const int tq = lane >> 3, q4 = 4 * (lane & 7);
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const int c4 = n * 32 + q4;
      const bool cok = c4 < p.N;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.b2 + (cok ? c4 : 0));
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = oacc[n][4 * g + e];
        *reinterpret_cast<f32x4*>(tr + l31 * 36 + 8 * g + 4 * hh) = v4;
      }
      f32x4 ov[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ov[i] = *reinterpret_cast<const f32x4*>(tr + (tq + 8 * i) * 36 + q4) + b4;
      if (cok) {
        float* op = p.out + (tok0 + tq) * p.ldo + c4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (tok0 + tq + 8 * i < p.M) *reinterpret_cast<f32x4*>(op + (long long)(8 * i) * p.ldo) = ov[i];
      }
    }
  }
  TM_T(41);
#ifdef TM_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TM_T(42);
#endif
}

// ---- plain-bf16 form of token_projmlp (nterms == 1) ----------------------------------------------------------------------------
// Same dataflow and operand orders as the split kernel above; what changes is the schedule (phase timings of the split form,
// tools/pm_time.cpp, 79.5 us per launch: att 8.1 | proj 6.4 | residuals 18.4 | LayerNorm 6.2 | MLP 29.3 = 12 x 2.44 | stores 6.2):
//   * no lo planes: the six projection tiles are ONE 78-KiB LDS image fetched at kernel start (no per-tile barrier), and the
//     MLP's W1 / W2 tiles are double buffered -- one barrier per hidden tile instead of two, each DMA a full tile ahead;
//   * the residual slice n+1 (x, conv) is in flight while slice n is transposed and tile n+1 is projected;
//   * gamma / beta come from LDS; GELU is x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3) -- |error| <= 4.8e-4, a sixteenth
//     of the bf16 rounding the value receives next -- 5 full-rate + 2 transcendental operations instead of 14 + 2.
#define PM_W1SLOT (13 * 1024)
#define PM_W2SLOT (15 * 1024)
#define PM_OFF_W2 (2 * PM_W1SLOT)
#define PM_PROJB (6 * PM_W1SLOT)
#define PM_PATCHB (32 * FF_XS_ROW * 4)
#define PM_OFF_PATCH PM_PROJB
#define PM_OFF_VEC (PM_OFF_PATCH + 8 * PM_PATCHB)

__global__ __launch_bounds__(512) void token_projmlp_bf16_kernel(TokenProjMlpParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Gs = reinterpret_cast<float*>(smem + PM_OFF_VEC);
  float* Bts = Gs + TM_KP;
  float* B1s = Bts + TM_KP;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const long long tok0 = (long long)blockIdx.x * 256 + wid * 32;
  float* patch = reinterpret_cast<float*>(smem + PM_OFF_PATCH + wid * PM_PATCHB);      // the wave's gather / transpose patch
  TM_T(0);

  // ---- the whole projection (6 tiles x 13 one-KiB pieces, hi plane) by LDS-DMA ----------------------------------------------
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int pc = wid + 8 * i;
    if (pc < 78) {
      const int tile = pc / 13, pw = pc - 13 * tile;
      int s = pw * 64 + lane;
      if (s > W1PL - 1) s = W1PL - 1;
      const int row = s / W1SLOTS;
      int q = s - row * W1SLOTS;
      if (q > 23) q = 23;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.wp + (long long)tile * (2 * TILE_ELEMS) + row * 192 + q * 8),
                                       (__attribute__((address_space(3))) void*)(smem + pc * 1024), 16, 0, 0);
    }
  }
  int off1[2], off2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int s = (wid + 8 * i) * 64 + lane;
    if (s > W1PL - 1) s = W1PL - 1;
    int row = s / W1SLOTS, q = s - row * W1SLOTS;
    if (q > 23) q = 23;
    off1[i] = row * 192 + q * 8;
    s = (wid + 8 * i) * 64 + lane;
    if (s > W2PL - 1) s = W2PL - 1;
    row = s / W2SLOTS; q = s - row * W2SLOTS;
    if (q > 3) q = 3;
    off2[i] = 2 * TILE_ELEMS + row * 32 + q * 8;
  }
  auto dma_mlp = [&](int ht, int slot) {
    const __bf16* rec = p.w + (long long)ht * (4 * TILE_ELEMS);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (wid + 8 * i < 13)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + off1[i]),
                                         (__attribute__((address_space(3))) void*)(smem + slot * PM_W1SLOT + (wid + 8 * i) * 1024), 16, 0, 0);
      if (wid + 8 * i < 15)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + off2[i]),
                                         (__attribute__((address_space(3))) void*)(smem + PM_OFF_W2 + slot * PM_W2SLOT + (wid + 8 * i) * 1024), 16, 0, 0);
    }
  };
  for (int i = tid; i < p.HT * 32; i += 512) B1s[i] = p.b1[i];
  if (tid < TM_KP) { Gs[tid] = tid < p.K ? p.gamma[tid] : 0.f; Bts[tid] = tid < p.K ? p.beta[tid] : 0.f; }

  // ---- att rows -> bf16 fragments -------------------------------------------------------------------------------------------
  bf16x8 xh[TM_KS];
  if (p.io_bf16 & 1) {
    // bf16 att rows (written by the attention kernels' bf16 stores): a lane's fragment of k-step st IS 16 bytes of its token's row
    const long long tk = tok0 + l31 < p.M ? tok0 + l31 : 0;
    const __bf16* arow = reinterpret_cast<const __bf16*>(p.att) + tk * p.lda + 8 * hh;
#pragma unroll
    for (int st = 0; st < TM_KS; ++st) {
      xh[st] = *reinterpret_cast<const bf16x8*>(arow + 16 * st);
      if (16 * st + 8 * hh + 7 >= p.K) {               // the row's padding is never written: mask it
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (16 * st + 8 * hh + j >= p.K) xh[st][j] = (__bf16)0.f;
      }
    }
  } else {
    float v[TM_KS][8];
    ff_wave_rows_to_frags<3>(p.att, p.lda, tok0, p.M, p.K, patch, lane, v);
#pragma unroll
    for (int st = 0; st < TM_KS; ++st)
#pragma unroll
      for (int j = 0; j < 8; ++j) xh[st][j] = (__bf16)v[st][j];
  }
  TM_T(1);

  // ---- residual slices: coalesced float4 loads (lane = token row lane>>3 (+8i), channel quad), one slice ahead ----------------
  const int tq = lane >> 3, q4 = 4 * (lane & 7);
  f32x4 rx[4], rc[4];
  // 32-bit element offsets of the lane's four token rows (launcher: M * ld < 2^31); rows past M read row 0 and channels past N
  // read channel 0 instead -- finite values that are never stored and that the zero-padded gamma / beta keep out of fc1
  int xo[4], co[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long tk = tok0 + tq + 8 * i;
    const int t = tk < p.M ? (int)tk : 0;
    xo[i] = t * p.ldx + q4;
    co[i] = t * p.ldc + q4;
  }
  auto load_res = [&](int n) {
    const int cn = (n * 32 + q4 < p.N) ? n * 32 : -q4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rx[i] = *reinterpret_cast<const f32x4*>(p.x + (xo[i] + cn));
      if (p.c2 && (p.io_bf16 & 2)) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const bf16x4 h4 = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.c2) + (co[i] + cn));
        rc[i] = (f32x4){(float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]};
      } else if (p.c2) rc[i] = *reinterpret_cast<const f32x4*>(p.c2 + (co[i] + cn));
    }
  };
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // the projection image, the vectors and the fragments are in place

  f32x16 oacc[6];
#pragma unroll
  for (int n = 0; n < 6; ++n) {
    const float* bpn = p.bp + n * 32 + 4 * hh;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[n][r] = bpn[(r & 3) + 8 * (r >> 2)];
    const unsigned char* ap = smem + n * PM_W1SLOT + l31 * W1ROWB + 16 * hh;
    bf16x8 fa[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) fa[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u);
#pragma unroll
    for (int st = 0; st < TM_KS; ++st) {
      const bf16x8 ah = fa[st & 1];
      if (st + 2 < TM_KS) fa[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2));
      __builtin_amdgcn_sched_barrier(0);
      oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[st], oacc[n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  TM_T(2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // every wave is past the projection image: the MLP's first two tiles may land on it
  TM_T(3);
  dma_mlp(0, 0);
  if (p.HT > 1) dma_mlp(1, 1);
#pragma unroll
  for (int n = 0; n < 6; ++n) {
    // x1 = proj + shortcut + conv * scale, slice n: through the wave's patch into accumulator order
    {
      load_res(n);
      const int c4 = n * 32 + q4;
      f32x4 sc = {0.f, 0.f, 0.f, 0.f};
      if (p.c2 && c4 < p.N) sc = *reinterpret_cast<const f32x4*>(p.rs2 + c4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 r0 = rx[i];
        if (p.c2) r0 += rc[i] * sc;
        *reinterpret_cast<f32x4*>(patch + (tq + 8 * i) * 36 + q4) = r0;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(patch + l31 * 36 + 8 * g + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) oacc[n][4 * g + e] += u[e];
      }
    }
  }
  TM_T(4);

  // ---- LayerNorm(x1) from the accumulators -> fc1 operand --------------------------------------------------------------------
  {
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += (n * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh < p.K) ? oacc[n][r] : 0.f;
    s += __shfl_xor(s, 32);
    const float mean = s / (float)p.K;
    float qv = 0.f;
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = (n * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh < p.K) ? oacc[n][r] - mean : 0.f;
        qv += d * d;
      }
    qv += __shfl_xor(qv, 32);
    const float rstd = 1.0f / sqrtf(qv / (float)p.K + p.eps);
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const int c0 = n * 32 + 8 * (2 * s2 + g2) + 4 * hh;           // channels of registers 8 s2 + 4 g2 .. + 3 (< 192: the vectors are zero padded)
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(Gs + c0);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(Bts + c0);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            xh[2 * n + s2][4 * g2 + e] = (__bf16)((oacc[n][8 * s2 + 4 * g2 + e] - mean) * rstd * g4[e] + b4[e]);
        }
        // the fragment is pinned here: left to itself the scheduler issues all 48 vector reads first and sinks the arithmetic
        // below them (192 live registers, 52 of them spilled)
        asm volatile("" : "+v"(xh[2 * n + s2]) :: "memory");
      }
  }
  TM_T(5);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // tiles 0 and 1 of the MLP have landed
  TM_T(6);

  // ---- the MLP over hidden tiles ---------------------------------------------------------------------------------------------
  for (int ht = 0; ht < p.HT; ++ht) {
    const int slot = ht & 1;
    f32x16 hacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = B1s[ht * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
    {
      const unsigned char* ap = smem + slot * PM_W1SLOT + l31 * W1ROWB + 16 * hh;
      bf16x8 fa[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) fa[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u);
#pragma unroll
      for (int st = 0; st < TM_KS; ++st) {
        const bf16x8 ah = fa[st & 1];
        if (st + 2 < TM_KS) fa[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2));
        __builtin_amdgcn_sched_barrier(0);
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[st], hacc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ht < 3) TM_T(8 + 4 * ht);
    bf16x8 gh[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const f32x2 g = ff_gelu_sig2((f32x2){hacc[8 * s + j], hacc[8 * s + j + 1]});
        gh[s][j] = (__bf16)g[0];
        gh[s][j + 1] = (__bf16)g[1];
      }
    if (ht < 3) TM_T(9 + 4 * ht);
    {
      const unsigned char* ap = smem + PM_OFF_W2 + slot * PM_W2SLOT + l31 * W2ROWB + 16 * hh;
      bf16x8 fa[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) fa[u] = *reinterpret_cast<const bf16x8*>(ap + (u >> 1) * 32 * W2ROWB + 32 * (u & 1));
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const int n = u >> 1, s2 = u & 1;
        const bf16x8 ah = fa[u & 1];
        if (u + 2 < 12) fa[u & 1] = *reinterpret_cast<const bf16x8*>(ap + ((u + 2) >> 1) * 32 * W2ROWB + 32 * ((u + 2) & 1));
        __builtin_amdgcn_sched_barrier(0);
        oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh[s2], oacc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ht < 3) TM_T(10 + 4 * ht);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // every wave is done with this slot; tile ht+1 has landed everywhere
    if (ht < 3) TM_T(11 + 4 * ht);
    if (ht + 2 < p.HT) dma_mlp(ht + 2, slot);
  }
  TM_T(40);

  // ---- epilogue: out = accumulators (x1 + fc2 part) + b2, transposed through the wave's patch into 128-byte row segments -----
  {
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const int c4 = n * 32 + q4;
      const bool cok = c4 < p.N;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.b2 + (cok ? c4 : 0));
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = oacc[n][4 * g + e];
        *reinterpret_cast<f32x4*>(patch + l31 * 36 + 8 * g + 4 * hh) = v4;
      }
      f32x4 ov[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ov[i] = *reinterpret_cast<const f32x4*>(patch + (tq + 8 * i) * 36 + q4) + b4;
      if (cok) {
        float* op = p.out + (tok0 + tq) * p.ldo + c4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (tok0 + tq + 8 * i < p.M) *reinterpret_cast<f32x4*>(op + (long long)(8 * i) * p.ldo) = ov[i];
      }
    }
  }
  TM_T(41);
#ifdef TM_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TM_T(42);
#endif
}

extern "C" int ff_token_projmlp(const float* att, int lda, const float* x, int ldx, const float* c2, int ldc, const float* c2_scale,
                                float* out, int ldo, long long M, int K, int hidden_tiles, const void* proj_tiles,
                                const float* proj_bias_padded, const float* gamma, const float* beta, float eps,
                                const void* mlp_tiles, const float* b1_padded, const float* b2, int nterms, int io_bf16, void* stream) {
  FF_CHECK_ARG(nterms == 1 || nterms == 3, "ff_token_projmlp: nterms must be 1 or 3");
  FF_CHECK_ARG(io_bf16 == 0 || (nterms == 1 && io_bf16 > 0 && io_bf16 < 4), "ff_token_projmlp: bf16 att / c2 rows exist for nterms == 1 only");
  FF_CHECK_ARG(!(io_bf16 & 1) || (lda % 8 == 0 && lda >= ((K + 15) / 16) * 16), "ff_token_projmlp: bf16 att rows must be 16-byte aligned and padded to a multiple of 16 channels");
  FF_CHECK_ARG(att && x && out && proj_tiles && proj_bias_padded && gamma && beta && mlp_tiles && b1_padded && b2, "ff_token_projmlp: null pointer");
  FF_CHECK_ARG(M > 0 && K > 0 && K <= TM_KP && K % 4 == 0 && hidden_tiles > 0, "ff_token_projmlp: needs K <= 192 (K %% 4 == 0)");
  FF_CHECK_ARG(lda >= K && lda % 4 == 0 && ldx >= K && ldx % 4 == 0 && ldo >= K && ldo % 4 == 0, "ff_token_projmlp: rows must be 16-byte aligned");
  FF_CHECK_ARG((((uintptr_t)att) & 15) == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)out) & 15) == 0, "ff_token_projmlp: 16-byte aligned tensors");
  FF_CHECK_ARG((c2 == nullptr) == (c2_scale == nullptr), "ff_token_projmlp: c2 / c2_scale come together");
  FF_CHECK_ARG(!c2 || (ldc >= K && ldc % 4 == 0 && (((uintptr_t)c2) & 15) == 0 && (((uintptr_t)c2_scale) & 15) == 0), "ff_token_projmlp: c2 rows must be 16-byte aligned");
  FF_CHECK_ARG((((uintptr_t)proj_tiles) & 15) == 0 && (((uintptr_t)mlp_tiles) & 15) == 0 && (((uintptr_t)gamma) & 15) == 0 &&
               (((uintptr_t)beta) & 15) == 0 && (((uintptr_t)b2) & 15) == 0, "ff_token_projmlp: weights / gamma / beta / b2 must be 16-byte aligned");
  TokenProjMlpParams p;
  p.att = att; p.x = x; p.c2 = c2; p.rs2 = c2_scale; p.out = out; p.gamma = gamma; p.beta = beta;
  p.wp = (const __bf16*)proj_tiles; p.bp = proj_bias_padded; p.w = (const __bf16*)mlp_tiles; p.b1 = b1_padded; p.b2 = b2;
  p.M = M; p.lda = lda; p.ldx = ldx; p.ldc = ldc; p.ldo = ldo; p.K = K; p.N = K; p.HT = hidden_tiles; p.eps = eps; p.io_bf16 = io_bf16;
#ifdef TM_TIMING
  p.dbg = g_tm_dbg;
#endif
  const size_t lds = (size_t)(2 * W1PL + 2 * W2PL) * 16 + (size_t)8 * 32 * XLROWB + (size_t)hidden_tiles * 32 * 4;
  FF_CHECK_ARG(lds <= 160 * 1024, "ff_token_projmlp: hidden too large for the LDS image");
  const long long nblk = (M + 255) / 256;
  FF_CHECK_ARG(nblk < (1LL << 31), "ff_token_projmlp: grid too large");
  FF_CHECK_ARG(M * (long long)(ldx > ldc ? ldx : ldc) < (1LL << 31), "ff_token_projmlp: too many tokens for 32-bit row offsets");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&token_projmlp_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&token_projmlp_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { ff_set_error("ff_token_projmlp: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; }
    attr_set = true;
  }
  static int v2 = -1;
  if (v2 < 0) { const char* e = getenv("FF_PM_V2"); v2 = (e && e[0] == '0') ? 0 : 1; }
  FF_CHECK_ARG(io_bf16 == 0 || v2, "ff_token_projmlp: bf16 att / c2 rows need the bf16 kernel (FF_PM_V2)");
  if (nterms == 1 && v2) {
    const size_t lds2 = (size_t)PM_OFF_VEC + (size_t)(2 * TM_KP + hidden_tiles * 32) * 4;
    FF_CHECK_ARG(lds2 <= 160 * 1024, "ff_token_projmlp: hidden too large for the LDS image");
    static bool attr2 = false;
    if (!attr2) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&token_projmlp_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { ff_set_error("ff_token_projmlp: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; }
      attr2 = true;
    }
    hipLaunchKernelGGL(token_projmlp_bf16_kernel, dim3((unsigned)nblk), dim3(512), lds2, (hipStream_t)stream, p);
  } else if (nterms == 3) hipLaunchKernelGGL(token_projmlp_kernel<3>, dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(token_projmlp_kernel<1>, dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, p);
  FF_LAUNCH_CHECK("ff_token_projmlp");
  return FF_OK;
}

extern "C" int ff_token_mlp(const float* x, int ldx, float* out, int ldo, long long M, int K, int hidden_tiles, int N,
                            const float* gamma, const float* beta, float eps, const void* w_tiles, const float* b1_padded,
                            const float* b2, int nterms, void* stream) {
  FF_CHECK_ARG(nterms == 1 || nterms == 3, "ff_token_mlp: nterms must be 1 or 3");
  FF_CHECK_ARG(x && out && gamma && beta && w_tiles && b1_padded && b2, "ff_token_mlp: null pointer");
  FF_CHECK_ARG(M > 0 && K > 0 && K <= TM_KP && K % 4 == 0 && N > 0 && N <= TM_NP && hidden_tiles > 0, "ff_token_mlp: needs K, N <= 192 (K %% 4 == 0)");
  FF_CHECK_ARG(ldx >= K && ldx % 4 == 0 && ldo >= N && (((uintptr_t)x) & 15) == 0, "ff_token_mlp: x rows must be 16-byte aligned");
  FF_CHECK_ARG((((uintptr_t)w_tiles) & 15) == 0 && (((uintptr_t)gamma) & 15) == 0 && (((uintptr_t)beta) & 15) == 0, "ff_token_mlp: weights / gamma / beta must be 16-byte aligned");
  FF_CHECK_ARG(ldx >= N, "ff_token_mlp: the residual is x itself, so N <= ldx");
  TokenMlpParams p;
  p.x = x; p.out = out; p.gamma = gamma; p.beta = beta; p.w = (const __bf16*)w_tiles;
  p.b1 = b1_padded; p.b2 = b2; p.M = M; p.ldx = ldx; p.ldo = ldo; p.K = K; p.N = N; p.HT = hidden_tiles; p.eps = eps;
  const bool vec4 = (N % 4 == 0) && (ldo % 4 == 0) && ((((uintptr_t)out) & 15) == 0) && ((((uintptr_t)b2) & 15) == 0);
  const size_t lds = (size_t)(2 * W1PL + 2 * W2PL) * 16 + (size_t)8 * 32 * XLROWB + (size_t)hidden_tiles * 32 * 4;
  FF_CHECK_ARG(lds <= 160 * 1024, "ff_token_mlp: hidden too large for the LDS image");
  const long long nblk = (M + 255) / 256;
  FF_CHECK_ARG(nblk < (1LL << 31), "ff_token_mlp: grid too large");
#define TM_LAUNCH(V4, NTM)                                                                                                 \
  do {                                                                                                                     \
    static bool attr_set = false;                                                                                          \
    if (!attr_set) {                                                                                                       \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&token_mlp_kernel<V4, NTM>),                        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                          \
      if (e != hipSuccess) { ff_set_error("ff_token_mlp: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; } \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    hipLaunchKernelGGL((token_mlp_kernel<V4, NTM>), dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, p);         \
  } while (0)
  if (nterms == 3) { if (vec4) TM_LAUNCH(true, 3); else TM_LAUNCH(false, 3); }
  else { if (vec4) TM_LAUNCH(true, 1); else TM_LAUNCH(false, 1); }
#undef TM_LAUNCH
  FF_LAUNCH_CHECK("ff_token_mlp");
  return FF_OK;
}

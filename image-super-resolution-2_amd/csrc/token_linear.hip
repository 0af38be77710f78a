// Token-stationary linear layer for the K <= 192 transformer projections:
//     out[token][n] = res[token][n] + res2[token][n]*rs2[n] + act( LayerNorm?(x)[token] . W[n] + bias[n] )
// (hat_arch.py:172 qkv after norm1 :272, :194 proj; dat_arch.py:501,559 qkv/proj, :163 fc1 after norm2 :735).
// These layers move 140-190 MB for 4-13 GFLOP at 65 536 tokens: they are HBM-bound, and the generic tiled GEMM
// (conv_gemm_bf16.hip) runs them latency-bound at 1.5 TB/s because K = 180 is only six 32-deep chunks per tile.
// Here a wave keeps its 32 tokens' (optionally layer-normalised) rows in REGISTERS as split-bf16 MFMA B operands
// for the whole kernel (x is read from HBM exactly once, LayerNorm costs no extra pass), the weight matrix streams
// through a two-slot LDS ring by LDS-DMA in 32-row tiles shared by the 8 waves, and every 32(n) x 32(token)
// accumulator tile is transposed through a wave-private LDS patch so stores are 128-byte row segments.
// Dataflow per n-tile: OUT^T[n][token] = W_tile . X^T   (A = W rows from LDS, B = X fragments), 36 MFMAs (bf16x3).
#include "ff_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KS = k-steps of 16 (K padded to 16*KS): 12 for the 180-wide transformers, 4 / 8 for NAFNet's 64 / 128 channels.
// Per W row: 2*KS data slots of 16 bytes + 1 pad slot; a tile (32 rows, hi + lo planes) is 2*32*(2KS+1)/64 DMA pieces
// (25 / 17 / 9 -- always integral).

struct TokenLinParams {
  const float* x; float* out;
  const float* gamma; const float* beta;   // NULL: no LayerNorm
  const __bf16* w;                         // [NT][2 planes][32][192], rows >= N zero
  const float* bias;                       // [NT*32] zero padded (or NULL)
  const float* res; const float* res2; const float* rs2;
  float* xn; int ldxn;                     // optional side output: the LayerNorm'ed rows (NULL: not written)
  float* stats; int st_lo, st_hi; float st_eps;   // optional side output: per token (mean, rstd) of OUTPUT channels [st_lo, st_hi)
  // GATED prologue (DAT adaptive interaction, dat_arch.py:541-556 / :649-664): the GEMM input is  x * cm[channel] + x2 * sm[token],
  // sm = sigmoid(gw2 . gelu(GW1 x + gb1) + gb2) computed from x itself (GW1: 180 -> 11, BatchNorm folded)
  const float* x2; int ldx2; const float* cm; const float* gw1t; const float* gb1; const float* gw2; float gb2;
  long long M;
  int ldx, ldo, ldr, ldr2, K, N, NT, act;
  float eps;
};

#define TL_TR 36   // floats per row of the transpose patch: 144 B keeps rows 16-byte aligned and ds_*_b128 conflict-free

// NTERMS = 3: split-bf16 products hi*lo + lo*hi + hi*hi (fp32-grade); NTERMS = 1: plain bf16 (hi*hi only: a third of the MFMAs, no lo
// fragments built or read; the weight image keeps its lo plane, unused).
#ifndef TL_GELU_SIG
#define TL_GELU_SIG 1
#endif
template <int ACT, int TL_KS, bool VEC4, bool GATED = false, int NTERMS = 3>
__global__ __launch_bounds__(512) void token_linear_kernel(TokenLinParams p) {
  constexpr int TL_SLOTS = 2 * TL_KS + 1, TL_ROWB = TL_SLOTS * 16, TL_PL = 32 * TL_SLOTS, TL_PIECES = 2 * TL_PL / 64;
  constexpr int TL_TILE_ELEMS = 32 * 16 * TL_KS, KPAD = 16 * TL_KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WB = TL_PL * 16;                     // bytes per plane
  constexpr int BUFB = 2 * WB;                       // one ring slot (hi + lo)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  float* tr = reinterpret_cast<float*>(smem + 2 * BUFB) + wid * (32 * TL_TR);   // wave-private transpose patch [token][36]
  float* xs = reinterpret_cast<float*>(smem + 2 * BUFB + 8 * 32 * TL_TR * 4) + wid * (32 * FF_XS_ROW);   // wave-private x staging
  float* Bs = reinterpret_cast<float*>(smem + 2 * BUFB + 8 * 32 * TL_TR * 4 + 8 * 32 * FF_XS_ROW * 4);   // [NT*32] bias
  const long long tok0 = (long long)blockIdx.x * 256 + wid * 32;
  const long long tok = tok0 + l31;
  const bool tvalid = tok < p.M;

  int off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int s = (wid + 8 * i) * 64 + lane;               // slot in the tile image (2 planes x 32 rows x 25 slots)
    if (s >= 2 * TL_PL) s = 2 * TL_PL - 1;
    const int plane = s / TL_PL, t = s - plane * TL_PL, row = t / TL_SLOTS;
    int q = t - row * TL_SLOTS;
    if (q > 2 * TL_KS - 1) q = 2 * TL_KS - 1;
    off[i] = plane * TL_TILE_ELEMS + row * KPAD + q * 8;
  }
  auto dma = [&](int nt, int buf) {
    const __bf16* rec = p.w + (long long)nt * (2 * TL_TILE_ELEMS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (wid + 8 * i < TL_PIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + off[i]),
                                         (__attribute__((address_space(3))) void*)(smem + buf * BUFB + (wid + 8 * i) * 1024), 16, 0, 0);
  };
  dma(0, 0);
  for (int i = tid; i < p.NT * 32; i += 512) Bs[i] = p.bias ? p.bias[i] : 0.f;

  // ---- x rows (+ LayerNorm) -> split bf16 fragments in registers ------------------------------------------------
  bf16x8 xh[TL_KS], xl[TL_KS];
  if constexpr (GATED) {
    // DAT's adaptive interaction folded into the projection's prologue: no pixel-gate kernel, no mix kernel, no 47 MB `fused`
    // tensor.  S = x supplies the per-token gate sm -- its 11-unit first layer is one more 32-row weight tile (index NT of the
    // image, rows 11..31 zero) against S's own fragments -- and takes the per-channel gate cm; O = x2 takes sm.
    dma(p.NT, 1);
    // Two phases, so that at most one set of fragments is live (the one-phase form -- S fragments kept through the O passes and
    // rewritten in place -- needed > 256 registers: 137 dwords of scratch per lane, ~70 MB of spill traffic per launch):
    //   phase 1: S rows -> temporary fragments -> gate GEMM -> sm (one float per lane pair);
    //   phase 2: S and O rows again, 64 channels per pass (S is an L2 hit now), combined in fp32 and split into the final fragments.
    float sm;
    {
      bf16x8 th[TL_KS], tl[TL_KS];
      {
        float v[TL_KS][8];
        ff_wave_rows_to_frags<TL_KS / 4>(p.x, p.ldx, tok0, p.M, p.K, xs, lane, v);
#pragma unroll
        for (int st = 0; st < TL_KS; ++st)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const __bf16 h = (__bf16)v[st][j];
            th[st][j] = h;
            if (NTERMS == 3) tl[st][j] = (__bf16)(v[st][j] - (float)h);
          }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                                         // tile 0 (slot 0) and the gate tile (slot 1) landed
      const unsigned char* ap = smem + BUFB + l31 * TL_ROWB + 16 * hh;
      f32x16 ga;
#pragma unroll
      for (int r = 0; r < 16; ++r) ga[r] = p.gb1[(r & 3) + 8 * (r >> 2) + 4 * hh];       // gb1 padded to 32
      bf16x8 fa[2], fl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        fa[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u);
        if (NTERMS == 3) fl[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u + WB);
      }
#pragma unroll
      for (int st = 0; st < TL_KS; ++st) {
        const bf16x8 ah = fa[st & 1], al = fl[st & 1];
        if (st + 2 < TL_KS) {
          fa[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2));
          if (NTERMS == 3) fl[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2) + WB);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NTERMS == 3) {
          ga = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, tl[st], ga, 0, 0, 0);
          ga = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, th[st], ga, 0, 0, 0);
        }
        ga = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, th[st], ga, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc += p.gw2[(r & 3) + 8 * (r >> 2) + 4 * hh] * ff_gelu_fast(ga[r]);   // gw2 padded with zeros
      sacc += __shfl_xor(sacc, 32);
      sm = 1.0f / (1.0f + expf(-(sacc + p.gb2)));
    }
    asm volatile("" ::: "memory");
    const int rr = lane >> 4, cq = (lane & 15) * 4;
#pragma unroll
    for (int pass = 0; pass < TL_KS / 4; ++pass) {
      asm volatile("" ::: "memory");                                           // keep the passes apart (no hoisting of all row loads)
      f32x4 tS[8], tO[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 4 * j + rr, c = 64 * pass + cq;
        const bool ok = tok0 + r < p.M && c < p.K;
        const f32x4 u = *reinterpret_cast<const f32x4*>(p.x + (ok ? (tok0 + r) * p.ldx + c : 0));
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(p.x2 + (ok ? (tok0 + r) * p.ldx2 + c : 0));
        tS[j] = ok ? u : (f32x4){0.f, 0.f, 0.f, 0.f};
        tO[j] = ok ? w2 : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(xs + (4 * j + rr) * FF_XS_ROW + cq) = tS[j];
      f32x4 sa[4], sb[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        sa[s4] = *reinterpret_cast<const f32x4*>(xs + l31 * FF_XS_ROW + 16 * s4 + 8 * hh);
        sb[s4] = *reinterpret_cast<const f32x4*>(xs + l31 * FF_XS_ROW + 16 * s4 + 8 * hh + 4);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(xs + (4 * j + rr) * FF_XS_ROW + cq) = tO[j];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int st = 4 * pass + s4, k0 = 16 * st + 8 * hh;
        const f32x4 oa = *reinterpret_cast<const f32x4*>(xs + l31 * FF_XS_ROW + 16 * s4 + 8 * hh);
        const f32x4 ob = *reinterpret_cast<const f32x4*>(xs + l31 * FF_XS_ROW + 16 * s4 + 8 * hh + 4);
        const int ka = k0 < p.K ? k0 : 0, kb = k0 + 4 < p.K ? k0 + 4 : 0;
        const f32x4 ca = *reinterpret_cast<const f32x4*>(p.cm + ka), cb = *reinterpret_cast<const f32x4*>(p.cm + kb);
        bf16x8 nh, nl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float o = j < 4 ? oa[j & 3] : ob[j & 3], cmv = j < 4 ? ca[j & 3] : cb[j & 3];
          const float sv = j < 4 ? sa[s4][j & 3] : sb[s4][j & 3];
          const float f = (k0 + j < p.K) ? sv * cmv + o * sm : 0.f;
          const __bf16 h = (__bf16)f;
          nh[j] = h;
          if (NTERMS == 3) nl[j] = (__bf16)(f - (float)h);
        }
        xh[st] = nh;
        if (NTERMS == 3) {
          xl[st] = nl;
          asm volatile("" : "+v"(xh[st]), "+v"(xl[st]) : : "memory");
        } else {
          asm volatile("" : "+v"(xh[st]) : : "memory");
        }            // the pass's fragments are FINISHED here: without this the
      }                                                                        // scheduler issues every pass's loads first and converts last (raw rows of all passes live)
    }
  } else
  {
    float v[TL_KS][8];
    ff_wave_rows_to_frags<TL_KS / 4>(p.x, p.ldx, tok0, p.M, p.K, xs, lane, v);
    float mean = 0.f, rstd = 1.f;
    if (p.gamma) {
      float s = 0.f;
#pragma unroll
      for (int st = 0; st < TL_KS; ++st)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[st][j];
      s += __shfl_xor(s, 32);
      mean = s / (float)p.K;
      float qv = 0.f;
#pragma unroll
      for (int st = 0; st < TL_KS; ++st)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = (16 * st + 8 * hh + j < p.K) ? v[st][j] - mean : 0.f;
          qv += d * d;
        }
      qv += __shfl_xor(qv, 32);
      rstd = 1.0f / sqrtf(qv / (float)p.K + p.eps);
    }
#pragma unroll
    for (int st = 0; st < TL_KS; ++st) {
      const int k0 = 16 * st + 8 * hh;
      f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (p.gamma) {
        const int ka = k0 < p.K ? k0 : 0, kb = k0 + 4 < p.K ? k0 + 4 : 0;
        g0 = *reinterpret_cast<const f32x4*>(p.gamma + ka); b0 = *reinterpret_cast<const f32x4*>(p.beta + ka);
        g1 = *reinterpret_cast<const f32x4*>(p.gamma + kb); b1 = *reinterpret_cast<const f32x4*>(p.beta + kb);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gg = j < 4 ? g0[j & 3] : g1[j & 3], bb = j < 4 ? b0[j & 3] : b1[j & 3];
        const float f = (k0 + j < p.K) ? (p.gamma ? (v[st][j] - mean) * rstd * gg + bb : v[st][j]) : 0.f;
        const __bf16 h = (__bf16)f;
        xh[st][j] = h;
        if (NTERMS == 3) xl[st][j] = (__bf16)(f - (float)h);
        v[st][j] = f;
      }
      // side output of the normalised rows (HAT feeds them to the CAB convolution, hat_arch.py:272-274): the separate
      // LayerNorm pass (read 47 MB, write 47 MB) becomes 47 MB of extra stores here.  A lane holds 8 consecutive channels
      // of its row: two float4 stores; the 12 steps of a wave complete whole lines in L2 before they are evicted.
      if (p.xn && tvalid) {
        float* xr = p.xn + tok * p.ldxn + k0;
        if (k0 < p.K) *reinterpret_cast<f32x4*>(xr) = (f32x4){v[st][0], v[st][1], v[st][2], v[st][3]};
        if (k0 + 4 < p.K) *reinterpret_cast<f32x4*>(xr + 4) = (f32x4){v[st][4], v[st][5], v[st][6], v[st][7]};
      }
    }
  }

  // LayerNorm statistics of a channel range of the (activated) OUTPUT, taken from the accumulators on their way out: the
  // consumer (DAT's SpatialGate: LayerNorm -> depth-wise 3x3, dat_arch.py:117-122) normalises on load, so the separate
  // LayerNorm pass over the 360-channel half (read 94 MB, write 94 MB) disappears
  float ssum = 0.f, ssq = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile 0 (this wave's DMA pieces) landed
  for (int nt = 0; nt < p.NT; ++nt) {
    const int buf = nt & 1;
    // raw barrier: every wave has waited for its pieces of tile nt (below, before its previous stores) and is done
    // reading ring slot buf^1.  No global-memory ordering is needed across it, so outstanding stores stay in flight.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nt + 1 < p.NT) dma(nt + 1, buf ^ 1);
    // residuals of this tile are fetched now and consumed after the MFMAs.  vec4 layout (rows 16-byte aligned, N % 4 == 0):
    // lane = (token row tq = lane >> 3 (+8 per step), channel quad cq4 = lane & 7) -> one float4 per lane and step, i.e.
    // 8 whole 128-byte row segments per instruction instead of 2 with dword accesses.
    const int col = nt * 32 + l31;
    const bool cok = col < p.N;
    const int cc = cok ? col : 0;
    const int tq = lane >> 3, c4 = nt * 32 + 4 * (lane & 7);
    const bool c4ok = c4 < p.N;
    float rv[16];
    f32x4 rq[4];
    if (VEC4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long tk = tok0 + tq + 8 * i;
        const bool ok = c4ok && tk < p.M;
        f32x4 r0 = {0.f, 0.f, 0.f, 0.f};
        if (p.res) { const f32x4 u = *reinterpret_cast<const f32x4*>(p.res + (ok ? tk * p.ldr + c4 : 0)); r0 = ok ? u : r0; }
        if (p.res2) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(p.res2 + (ok ? tk * p.ldr2 + c4 : 0));
          const f32x4 sc = *reinterpret_cast<const f32x4*>(p.rs2 + (c4ok ? c4 : 0));
          if (ok) r0 += u * sc;
        }
        rq[i] = r0;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const long long tk = tok0 + 2 * i + hh;
        const long long tqq = tk < p.M ? tk : 0;
        float r0 = p.res ? p.res[tqq * p.ldr + cc] : 0.f;
        if (p.res2) r0 += p.res2[tqq * p.ldr2 + cc] * p.rs2[cc];
        rv[i] = r0;
      }
    }
    const unsigned char* ap = smem + buf * BUFB + l31 * TL_ROWB + 16 * hh;
    f32x16 acc;
    {
      const int nb = nt * 32 + 4 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = Bs[nb + (r & 3) + 8 * (r >> 2)];
    }
    bf16x8 fa[2], fl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      fa[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u);
      if (NTERMS == 3) fl[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u + WB);
    }
#pragma unroll
    for (int st = 0; st < TL_KS; ++st) {
      const bf16x8 ah = fa[st & 1], al = fl[st & 1];
      if (st + 2 < TL_KS) {
        fa[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2));
        if (NTERMS == 3) fl[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2) + WB);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (NTERMS == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl[st], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[st], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[st], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue of this tile: act, transpose through the wave's LDS patch, coalesced row-segment stores --------
    // accumulator register r of lane (l31 = token, hh) is channel (r & 3) + 8 (r >> 2) + 4 hh: four consecutive channels
    // per r >> 2 -> one ds_write_b128 each
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v4;
      if (ACT == ACT_GELU && NTERMS == 1 && TL_GELU_SIG) {      // plain bf16: the sigmoid form (ff_common.h)
        const f32x2 ga = ff_gelu_sig2((f32x2){acc[4 * g], acc[4 * g + 1]}), gb = ff_gelu_sig2((f32x2){acc[4 * g + 2], acc[4 * g + 3]});
        v4 = (f32x4){ga[0], ga[1], gb[0], gb[1]};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = ff_act_c<ACT, true>(acc[4 * g + e]);
      }
      *reinterpret_cast<f32x4*>(tr + l31 * TL_TR + 8 * g + 4 * hh) = v4;
      if (p.stats) {
        const int c0 = nt * 32 + 8 * g + 4 * hh;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c0 + e >= p.st_lo && c0 + e < p.st_hi) { ssum += v4[e]; ssq += v4[e] * v4[e]; }
      }
    }
    // One wait per tile, placed BEFORE this tile's stores: it retires the DMA of tile nt+1 (issued a whole tile ago),
    // this tile's residual loads and the PREVIOUS tile's stores -- so stores always have a full tile to drain
    // (vmcnt counts stores too; waiting right after issuing them would serialise the kernel on HBM write latency).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (VEC4) {
      f32x4 ov[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ov[i] = *reinterpret_cast<const f32x4*>(tr + (tq + 8 * i) * TL_TR + 4 * (lane & 7)) + rq[i];
      if (c4ok) {
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
    } else {
      float ov[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) ov[i] = tr[(2 * i + hh) * TL_TR + l31] + rv[i];      // 16 LDS reads back to back
      if (cok) {                                                                        // one exec mask for the whole tile
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
  if (p.stats) {
    ssum += __shfl_xor(ssum, 32);
    ssq += __shfl_xor(ssq, 32);
    const float n = (float)(p.st_hi - p.st_lo);
    const float mean = ssum / n;
    const float var = fmaxf(ssq / n - mean * mean, 0.f);
    if (hh == 0 && tvalid) *reinterpret_cast<float2*>(p.stats + 2 * tok) = make_float2(mean, 1.0f / sqrtf(var + p.st_eps));
  }
}


extern "C" int ff_token_linear(const float* x, int ldx, float* out, int ldo, long long M, int K, int kpad, int N, int n_tiles,
                               const float* gamma, const float* beta, float eps, const void* w_tiles,
                               const float* bias_padded, int act, const float* res, int ldr, const float* res2, int ldr2,
                               const float* res2_scale, float* xn_out, int ldxn, float* stats_out, int stat_lo, int stat_hi,
                               float stat_eps, int nterms, void* stream) {
  FF_CHECK_ARG(nterms == 1 || nterms == 3, "ff_token_linear: nterms must be 1 (plain bf16) or 3 (split bf16)");
  FF_CHECK_ARG(x && out && w_tiles, "ff_token_linear: null pointer");
  FF_CHECK_ARG(M > 0 && K > 0 && K <= 192 && K % 4 == 0 && N > 0 && n_tiles * 32 >= N, "ff_token_linear: needs K <= 192 (K %% 4 == 0), n_tiles*32 >= N");
  FF_CHECK_ARG(kpad == 64 || kpad == 128 || kpad == 192, "ff_token_linear: kpad must be 64, 128 or 192");
  FF_CHECK_ARG(kpad >= K, "ff_token_linear: kpad < K");
  FF_CHECK_ARG(ldx >= K && ldx % 4 == 0 && ldo >= N && (((uintptr_t)x) & 15) == 0, "ff_token_linear: x rows must be 16-byte aligned");
  FF_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "ff_token_linear: gamma/beta come together");
  FF_CHECK_ARG((((uintptr_t)w_tiles) & 15) == 0 && (!gamma || ((((uintptr_t)gamma) & 15) == 0 && (((uintptr_t)beta) & 15) == 0)), "ff_token_linear: weights / gamma / beta must be 16-byte aligned");
  FF_CHECK_ARG(!res || ldr >= N, "ff_token_linear: ldr too small");
  FF_CHECK_ARG(!res2 || (ldr2 >= N && res2_scale), "ff_token_linear: res2 needs ldr2 >= N and a scale vector");
  FF_CHECK_ARG(!xn_out || (gamma && ldxn >= K && ldxn % 4 == 0 && (((uintptr_t)xn_out) & 15) == 0), "ff_token_linear: xn_out needs LayerNorm parameters and 16-byte aligned rows");
  TokenLinParams p;
  p.x = x; p.out = out; p.gamma = gamma; p.beta = beta; p.w = (const __bf16*)w_tiles; p.bias = bias_padded;
  p.res = res; p.res2 = res2; p.rs2 = res2_scale; p.xn = xn_out; p.ldxn = ldxn; p.M = M; p.ldx = ldx; p.ldo = ldo; p.ldr = ldr; p.ldr2 = ldr2;
  p.K = K; p.N = N; p.NT = n_tiles; p.act = act; p.eps = eps;
  FF_CHECK_ARG(!stats_out || (stat_lo >= 0 && stat_hi > stat_lo && stat_hi <= N && (((uintptr_t)stats_out) & 7) == 0), "ff_token_linear: bad statistics range");
  p.stats = stats_out; p.st_lo = stat_lo; p.st_hi = stat_hi; p.st_eps = stat_eps;
  p.x2 = nullptr; p.ldx2 = 0; p.cm = nullptr; p.gw1t = nullptr; p.gb1 = nullptr; p.gw2 = nullptr; p.gb2 = 0.f;
  const bool vec4 = (N % 4 == 0) && (ldo % 4 == 0) && ((((uintptr_t)out) & 15) == 0) && (!res || (ldr % 4 == 0 && (((uintptr_t)res) & 15) == 0)) &&
           (!res2 || (ldr2 % 4 == 0 && (((uintptr_t)res2) & 15) == 0 && (((uintptr_t)res2_scale) & 15) == 0));
  const int ks = kpad / 16;
  const size_t lds = (size_t)2 * 2 * 32 * (2 * ks + 1) * 16 + (size_t)8 * 32 * TL_TR * 4 + (size_t)8 * 32 * FF_XS_ROW * 4 + (size_t)n_tiles * 32 * 4;
  FF_CHECK_ARG(lds <= 160 * 1024, "ff_token_linear: N too large for the LDS image");
  FF_CHECK_ARG(act == ACT_NONE || act == ACT_GELU, "ff_token_linear: act must be none or gelu");
  const long long nblk = (M + 255) / 256;
  FF_CHECK_ARG(nblk < (1LL << 31), "ff_token_linear: grid too large");
#define TL_LAUNCH(A, KSV, V4, NTM)                                                                                          \
  do {                                                                                                                      \
    static bool attr_set = false;                                                                                           \
    if (!attr_set) {                                                                                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&token_linear_kernel<A, KSV, V4, false, NTM>),       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                           \
      if (e != hipSuccess) { ff_set_error("ff_token_linear: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; } \
      attr_set = true;                                                                                                      \
    }                                                                                                                       \
    hipLaunchKernelGGL((token_linear_kernel<A, KSV, V4, false, NTM>), dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, p); \
  } while (0)
#define TL_LAUNCH_NT(A, KSV, V4) do { if (nterms == 3) TL_LAUNCH(A, KSV, V4, 3); else TL_LAUNCH(A, KSV, V4, 1); } while (0)
#define TL_LAUNCH_KS(A, V4) do { if (ks == 4) TL_LAUNCH_NT(A, 4, V4); else if (ks == 8) TL_LAUNCH_NT(A, 8, V4); else TL_LAUNCH_NT(A, 12, V4); } while (0)
  if (act == ACT_GELU) { if (vec4) TL_LAUNCH_KS(ACT_GELU, true); else TL_LAUNCH_KS(ACT_GELU, false); }
  else { if (vec4) TL_LAUNCH_KS(ACT_NONE, true); else TL_LAUNCH_KS(ACT_NONE, false); }
#undef TL_LAUNCH_KS
#undef TL_LAUNCH_NT
#undef TL_LAUNCH
  FF_LAUNCH_CHECK("ff_token_linear");
  return FF_OK;
}

// out = res + W . ( x * cm[channel] + x2 * sm[token] ) + b,  sm = sigmoid(gw2 . gelu(GW1 x + gb1) + gb2):  DAT's adaptive interaction
// (channel gate on one branch, spatial gate on the other, dat_arch.py:541-556 spatial blocks / :649-664 channel blocks) + the output
// projection (:559 / :666) + the block's residual in one launch.  gw1t: [192][12] floats = GW1 transposed and zero padded
// (11 hidden units, BatchNorm folded), gb1 / gw2: [12].
extern "C" int ff_token_linear_gated(const float* x, int ldx, const float* x2, int ldx2, const float* cm, const float* gw1t,
                                     const float* gb1, const float* gw2, float gb2, float* out, int ldo, long long M, int K, int N,
                                     int n_tiles, const void* w_tiles, const float* bias_padded, const float* res, int ldr, int nterms,
                                     void* stream) {
  FF_CHECK_ARG(nterms == 1 || nterms == 3, "ff_token_linear_gated: nterms must be 1 or 3");
  FF_CHECK_ARG(x && x2 && cm && gb1 && gw2 && out && w_tiles, "ff_token_linear_gated: null pointer");   /* gw1t is unused: GW1 travels as tile n_tiles of w_tiles */
  FF_CHECK_ARG(M > 0 && K > 0 && K <= 192 && K % 4 == 0 && N > 0 && N % 4 == 0 && n_tiles * 32 >= N, "ff_token_linear_gated: needs K <= 192, K and N multiples of 4");
  FF_CHECK_ARG(ldx >= K && ldx % 4 == 0 && ldx2 >= K && ldx2 % 4 == 0 && ldo >= N && ldo % 4 == 0 && (!res || (ldr >= N && ldr % 4 == 0)), "ff_token_linear_gated: rows must be 16-byte aligned");
  FF_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)x2) | ((uintptr_t)cm) | ((uintptr_t)out) | ((uintptr_t)w_tiles) | ((uintptr_t)res)) & 15) == 0, "ff_token_linear_gated: 16-byte alignment");
  TokenLinParams p;
  p.x = x; p.out = out; p.gamma = nullptr; p.beta = nullptr; p.w = (const __bf16*)w_tiles; p.bias = bias_padded;
  p.res = res; p.res2 = nullptr; p.rs2 = nullptr; p.xn = nullptr; p.ldxn = 0; p.M = M; p.ldx = ldx; p.ldo = ldo; p.ldr = ldr; p.ldr2 = 0;
  p.K = K; p.N = N; p.NT = n_tiles; p.act = ACT_NONE; p.eps = 0.f;
  p.stats = nullptr; p.st_lo = 0; p.st_hi = 0; p.st_eps = 0.f;
  p.x2 = x2; p.ldx2 = ldx2; p.cm = cm; p.gw1t = gw1t; p.gb1 = gb1; p.gw2 = gw2; p.gb2 = gb2;
  const size_t lds = (size_t)2 * 2 * 32 * 25 * 16 + (size_t)8 * 32 * TL_TR * 4 + (size_t)8 * 32 * FF_XS_ROW * 4 + (size_t)n_tiles * 32 * 4;
  FF_CHECK_ARG(lds <= 160 * 1024, "ff_token_linear_gated: N too large for the LDS image");
  const long long nblk = (M + 255) / 256;
  FF_CHECK_ARG(nblk < (1LL << 31), "ff_token_linear_gated: grid too large");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&token_linear_kernel<ACT_NONE, 12, true, true, 3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&token_linear_kernel<ACT_NONE, 12, true, true, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { ff_set_error("ff_token_linear_gated: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; }
    attr_set = true;
  }
  if (nterms == 3) hipLaunchKernelGGL((token_linear_kernel<ACT_NONE, 12, true, true, 3>), dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((token_linear_kernel<ACT_NONE, 12, true, true, 1>), dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, p);
  FF_LAUNCH_CHECK("ff_token_linear_gated");
  return FF_OK;
}

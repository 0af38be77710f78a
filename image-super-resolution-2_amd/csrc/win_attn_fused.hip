// Window-resident attention block: LayerNorm -> q/k/v projection -> softmax(q k^T + bias (+ shift mask)) v, one workgroup per
// window handling ALL its heads, so the 141 MB qkv tensor of the two-stage form (token_linear -> window_attn) never exists.
//   HAT  W-MSA / SW-MSA   hat_arch.py:165-196 (WindowAttention.forward), :266-309 (HAB: norm1, roll, window partition)
//   DAT  spatial branches dat_arch.py:290-342 (SpatialAttention.forward), :491-562 (qkv, zero pad to x32, shift, DynamicPosBias)
// Dataflow of a 512-thread workgroup (8 waves, two per SIMD; wave w owns window tokens 32w..32w+31 as queries AND as keys):
//   prologue   the wave gathers its 32 token rows of x (cyclic shift = index arithmetic), LayerNorm in registers, rows kept
//              as split-bf16 MFMA fragments (96 VGPRs) for the whole kernel; the normalised rows are an optional side output
//              (HAT's CAB branch reads them);
//   per head   three 32-row weight tiles (q_h, k_h, v_h; head dim padded 30 -> 32, softmax scale folded into q) stream
//              through a 2-slot LDS ring by LDS-DMA, 36 MFMAs each (bf16x3):
//                q^T = Wq . x^T      stays in registers: its accumulator registers 8s..8s+7 ARE the B operand of k-step s
//                k^T = Wk . x^T      lane = key, 16 channels per lane: two 16-byte LDS stores per plane into K[key][d-permuted]
//                v   = x . Wv^T      operands swapped so the lane is the channel: two 16-byte stores per plane into V^T[d][key]
//              then S^T = K Q^T + bias over 64-key chunks (online softmax), P^T from the accumulators, O^T += V^T P^T;
//              the relative-position bias is gathered from the head's COMPACT (2wh-1) x (2ww-1) table in LDS (6-10 KB,
//              row stride chosen bank-conflict free) with compile-time offsets -- no expanded [keys][queries] table;
//              O^T is transposed through the wave's own (now idle) K rows and stored as 120-byte row segments.
// HBM traffic per launch: x in, attention out (+ the optional side output): 2-3 x 47 MB instead of 47+141+47 | 141+180+47.
#include "ff_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct WinFusedParams {
  const float* x; float* out;
  const float* gamma; const float* beta;
  const __bf16* w;        // [3*heads_total tiles][2 planes][32][192]  tile 3g+{0,1,2} = q,k,v rows of head g
  const float* bias;      // [3*heads_total*32] zero padded
  const float* rel;       // [heads_total][rel_rows][rel_stride] compact relative-position bias, row padded
  float* xn; float* vout;
  int ldx, ldo, ldxn, ldv, o_off, v_off;
  int B, H, W, Hp, Wp, sh, sw, use_mask, nwx, nwy;
  int head0, nheads, d, K, zero_pad, rel_rows, rel_stride, prio;
  int out_bf16, xn_bf16;   // plain-bf16 kernel only: out / xn are bf16 rows (ldo / ldxn in elements) -- their consumers round to bf16 anyway
  float eps;
#ifdef WF_TIMING
  unsigned long long* dbg;   // tools/wf_time.cpp: [block][wave][64] wall-clock stamps (debug build only)
#endif
};
#ifdef WF_TIMING
static unsigned long long* g_wf_dbg = nullptr;
#define WF_T(i) do { if (p.dbg && lane == 0) p.dbg[((long long)blockIdx.x * 8 + wid) * 64 + (i)] = wall_clock64(); } while (0)
#else
#define WF_T(i) do { } while (0)
#endif

#define WF_KS 12
#define WF_SLOTS 25                        // 16-byte slots per weight row in LDS (24 data + 1 pad)
#define WF_ROWB (WF_SLOTS * 16)
#define WF_PL (32 * WF_SLOTS)              // slots per plane
#define WF_PLB (WF_PL * 16)                // bytes per plane
#define WF_BUFB (2 * WF_PLB)               // one ring slot (hi + lo) = 25 600 B = 25 DMA pieces
#define WF_TILE_ELEMS (32 * 192)
#define WF_KROWB 80
#define WF_KWAVE (2 * 32 * WF_KROWB)       // a wave's K rows, both planes: 5120 B (also its output transpose patch)
#define WF_VROWB 528
#define WF_VPLB (32 * WF_VROWB)
#define WF_OFF_K (2 * WF_BUFB)
#define WF_OFF_V (WF_OFF_K + 8 * WF_KWAVE)
#define WF_OFF_REL (WF_OFF_V + 2 * WF_VPLB)
#define WF_REL_MAX 2560                    // floats per head table (63 x 40 for the 32x8 window is the largest)
#define WF_REL_PAD 4                       // leading pad: keeps the lane part of the index non-negative
#define WF_OFF_BS (WF_OFF_REL + (WF_REL_MAX + WF_REL_PAD) * 4)
#define WF_BS_MAX 576
#define WF_OFF_TOK (WF_OFF_BS + WF_BS_MAX * 4)
#define WF_OFF_REG (WF_OFF_TOK + 1024)
#define WF_OFF_SAME (WF_OFF_REG + 1024)     // [9 regions][8 key tiles] 32-bit masks: which keys of the tile lie in the region
#define WF_LDS (WF_OFF_SAME + 9 * 8 * 4)
#define WF_XS_ROW 68

template <int WW, int NTERMS, int UNR>
__global__ __launch_bounds__(512) void win_attn_fused_kernel(WinFusedParams p) {
  constexpr int WH = 256 / WW;
  // row stride of the compact bias table: >= 2 WW - 1 and == WW (mod 32), so the WH' rows of queries a 32-lane group spans
  // fall on disjoint banks (8 -> 40, 16 -> 48, 32 -> 64)
  constexpr int RS = WW == 8 ? 40 : (WW == 16 ? 48 : 64);
  constexpr int CMAX = (WH - 1) * RS + WW - 1;          // largest key part of the index
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* KB = smem + WF_OFF_K;
  unsigned char* VB = smem + WF_OFF_V;
  float* REL = reinterpret_cast<float*>(smem + WF_OFF_REL);
  float* Bs = reinterpret_cast<float*>(smem + WF_OFF_BS);
  int* ktok = reinterpret_cast<int*>(smem + WF_OFF_TOK);
  int* kreg = reinterpret_cast<int*>(smem + WF_OFF_REG);
  unsigned* same = reinterpret_cast<unsigned*>(smem + WF_OFF_SAME);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  int bid = blockIdx.x;
  const int wx = bid % p.nwx; bid /= p.nwx;
  const int wy = bid % p.nwy;
  const int b = bid / p.nwy;
  WF_T(0);

  // ---- weight-tile DMA: piece (wid + 8 i) of the 25 one-KiB pieces of a tile image --------------------------------------
  int off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int s = (wid + 8 * i) * 64 + lane;
    if (s >= 2 * WF_PL) s = 2 * WF_PL - 1;
    const int plane = s / WF_PL, t = s - plane * WF_PL, row = t / WF_SLOTS;
    int q = t - row * WF_SLOTS;
    if (q > 2 * WF_KS - 1) q = 2 * WF_KS - 1;
    off[i] = plane * WF_TILE_ELEMS + row * 192 + q * 8;
  }
  auto dma = [&](int tile, int buf) {
    const __bf16* rec = p.w + (long long)tile * (2 * WF_TILE_ELEMS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (wid + 8 * i < 25)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + off[i]),
                                         (__attribute__((address_space(3))) void*)(smem + buf * WF_BUFB + (wid + 8 * i) * 1024), 16, 0, 0);
  };
  const int tile0 = 3 * p.head0, ntiles = 3 * p.nheads;
  dma(tile0, 0);

  // ---- window tables: token index of every window position (after the cyclic shift; -1 outside the image) and its
  //      shift-mask region; the padded bias vector of this head group --------------------------------------------------
  if (tid < 256) {
    const int qy = wy * WH + tid / WW, qx = wx * WW + tid % WW;
    int oy = qy + p.sh, ox = qx + p.sw;
    if (oy >= p.Hp) oy -= p.Hp;
    if (ox >= p.Wp) ox -= p.Wp;
    ktok[tid] = (oy < p.H && ox < p.W) ? (b * p.H + oy) * p.W + ox : -1;
    int rid = 0;
    if (p.use_mask) {
      const int ry = qy < p.Hp - WH ? 0 : (qy < p.Hp - p.sh ? 1 : 2);
      const int rx = qx < p.Wp - WW ? 0 : (qx < p.Wp - p.sw ? 1 : 2);
      rid = 3 * ry + rx;
    }
    kreg[tid] = rid;
  }
  for (int i = tid; i < ntiles * 32; i += 512) Bs[i] = p.bias ? p.bias[tile0 * 32 + i] : 0.f;
  __syncthreads();

  const bool blk_mask = p.use_mask && (wy == p.nwy - 1 || wx == p.nwx - 1);
  if (blk_mask) {                                      // key-region bit masks: the shift mask becomes one LDS word per key tile
    for (int i = wid; i < 72; i += 8) {
      const int region = i >> 3, t = i & 7;
      const unsigned long long bal = __ballot(l31 == lane && kreg[32 * t + l31] == region);
      if (lane == 0) same[i] = (unsigned)bal;
    }
    __syncthreads();
  }
  const int qi = wid * 32 + l31;                       // this lane's window position (query; also one of the wave's keys)
  const int mytok = ktok[qi];
  const bool tvalid = mytok >= 0;
  const int qreg_id = kreg[qi];
  // validity of the wave's 32 tokens as a bit mask (the v tile has tokens in its registers, not in its lanes)
  const unsigned vmask = (unsigned)(__ballot(tvalid) & 0xffffffffu);
  const bool kill_pad = p.zero_pad != 0;               // DAT: q/k/v of the zero-padded tokens are exactly zero
  WF_T(1);

  // ---- the wave's 32 rows of x -> LayerNorm -> split-bf16 fragments ---------------------------------------------------
  bf16x8 xh[WF_KS], xl[WF_KS];
  {
    float* xs = reinterpret_cast<float*>(smem + WF_OFF_K) + wid * (32 * WF_XS_ROW);    // gather patch over the idle K/V area
    const int rr = lane >> 4, cq = (lane & 15) * 4;
    float v[WF_KS][8];
    f32x4 t[3][8];
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 4 * j + rr, c = 64 * pass + cq;
        const int tk = ktok[wid * 32 + r];
        const bool ok = tk >= 0 && c < p.K;
        const f32x4 u = *reinterpret_cast<const f32x4*>(p.x + (ok ? (long long)tk * p.ldx + c : 0));
        t[pass][j] = ok ? u : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(xs + (4 * j + rr) * WF_XS_ROW + cq) = t[pass][j];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xs + l31 * WF_XS_ROW + 16 * s4 + 8 * hh);
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(xs + l31 * WF_XS_ROW + 16 * s4 + 8 * hh + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[4 * pass + s4][e] = a[e]; v[4 * pass + s4][4 + e] = c4[e]; }
      }
    }
    float mean = 0.f, rstd = 1.f;
    if (p.gamma) {
      float s = 0.f;
#pragma unroll
      for (int st = 0; st < WF_KS; ++st)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[st][j];
      s += __shfl_xor(s, 32);
      mean = s / (float)p.K;
      float qv = 0.f;
#pragma unroll
      for (int st = 0; st < WF_KS; ++st)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dd = (16 * st + 8 * hh + j < p.K) ? v[st][j] - mean : 0.f;
          qv += dd * dd;
        }
      qv += __shfl_xor(qv, 32);
      rstd = 1.0f / sqrtf(qv / (float)p.K + p.eps);
    }
#pragma unroll
    for (int st = 0; st < WF_KS; ++st) {
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
      if (p.xn && tvalid) {                          // side output: the normalised rows (two float4 per step and lane)
        float* xr = p.xn + (long long)mytok * p.ldxn + k0;
        if (k0 < p.K) *reinterpret_cast<f32x4*>(xr) = (f32x4){v[st][0], v[st][1], v[st][2], v[st][3]};
        if (k0 + 4 < p.K) *reinterpret_cast<f32x4*>(xr + 4) = (f32x4){v[st][4], v[st][5], v[st][6], v[st][7]};
      }
    }
  }

  WF_T(2);
  // lane part of the relative-position index: (qy + WH-1) * stride + qx + WW-1 - 4 hh   (window-local coordinates)
  // bias[q][k] = T[(qy - ky + WH-1) * RS + (qx - kx + WW-1)]: lane part minus a compile-time key part; written as
  // REL[rel_base + (CMAX - key part)] so every offset is a non-negative immediate of the ds_read
  const float* rel_base = REL + WF_REL_PAD + ((qi / WW) + WH - 1) * RS + (qi % WW) + WW - 1 - 4 * hh - CMAX;
  const float LOG2E = 1.4426950408889634f;
  unsigned char* kmine = KB + wid * WF_KWAVE + l31 * WF_KROWB + 16 * hh;
  int it = 0;                                           // running tile counter: ring slot = it & 1

  // one 32-row weight tile times the wave's tokens.  SWAP = false: acc[ch][token] (lane = token); true: acc[token][ch]
  auto gemm_tile = [&](int buf, bool swap, f32x16& acc) {
    const unsigned char* ap = smem + buf * WF_BUFB + l31 * WF_ROWB + 16 * hh;
    bf16x8 fa[2], fl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      fa[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u);
      if (NTERMS == 3) fl[u] = *reinterpret_cast<const bf16x8*>(ap + 32 * u + WF_PLB);
    }
#pragma unroll
    for (int st = 0; st < WF_KS; ++st) {
      const bf16x8 ah = fa[st & 1], al = fl[st & 1];
      if (st + 2 < WF_KS) {
        fa[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2));
        if (NTERMS == 3) fl[st & 1] = *reinterpret_cast<const bf16x8*>(ap + 32 * (st + 2) + WF_PLB);
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
  // ring hand-over: own DMA pieces of tile `it` have landed, every wave is past its reads of the other slot
  auto ring_step = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 1 < ntiles) dma(tile0 + it + 1, (it + 1) & 1);
  };

  if (p.prio && wid >= 4) __builtin_amdgcn_s_setprio(1);       // static priority for the younger half (guide: two waves per SIMD, item 4)
  for (int hi = 0; hi < p.nheads; ++hi) {
    const int g = p.head0 + hi;
    // this head's compact bias table -> LDS (every wave is past the previous head's attention: barrier at the loop end)
    {
      const float* src = p.rel + (long long)g * p.rel_rows * p.rel_stride;
      for (int i = tid; i < p.rel_rows * p.rel_stride; i += 512) REL[WF_REL_PAD + i] = src[i];
    }
    // ---- q tile ----------------------------------------------------------------------------------------------------
    bf16x8 qh[2], ql[2];
    {
      ring_step();
      WF_T(3 + 8 * hi);
      f32x16 acc;
      const int nb = it * 32 + 4 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = Bs[nb + (r & 3) + 8 * (r >> 2)];
      gemm_tile(it & 1, false, acc);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = acc[8 * s + j];
          const __bf16 h = (__bf16)f;
          qh[s][j] = h;
          if (NTERMS == 3) ql[s][j] = (__bf16)(f - (float)h);
        }
      ++it;
    }
    // ---- k tile -> K[key][d-permuted] --------------------------------------------------------------------------------
    {
      WF_T(4 + 8 * hi);
      ring_step();
      WF_T(5 + 8 * hi);
      f32x16 acc;
      const int nb = it * 32 + 4 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = Bs[nb + (r & 3) + 8 * (r >> 2)];
      gemm_tile(it & 1, false, acc);
      const bool zero = kill_pad && !tvalid;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 h8, l8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = zero ? 0.f : acc[8 * s + j];
          const __bf16 h = (__bf16)f;
          h8[j] = h;
          if (NTERMS == 3) l8[j] = (__bf16)(f - (float)h);
        }
        *reinterpret_cast<bf16x8*>(kmine + 32 * s) = h8;
        if (NTERMS == 3) *reinterpret_cast<bf16x8*>(kmine + 32 * s + 32 * WF_KROWB) = l8;
      }
      ++it;
    }
    // ---- v tile (operands swapped: lane = channel) -> V^T[d][key-permuted] (+ optional side output) ------------------
    {
      WF_T(6 + 8 * hi);
      ring_step();
      f32x16 acc;
      const float bv = Bs[it * 32 + l31];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bv;
      gemm_tile(it & 1, true, acc);
      if (kill_pad) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!((vmask >> ((r & 3) + 8 * (r >> 2) + 4 * hh)) & 1u)) acc[r] = 0.f;
      }
      unsigned char* vp = VB + l31 * WF_VROWB + 64 * wid + 16 * hh;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 h8, l8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = acc[8 * s + j];
          const __bf16 h = (__bf16)f;
          h8[j] = h;
          if (NTERMS == 3) l8[j] = (__bf16)(f - (float)h);
        }
        *reinterpret_cast<bf16x8*>(vp + 32 * s) = h8;
        if (NTERMS == 3) *reinterpret_cast<bf16x8*>(vp + 32 * s + WF_VPLB) = l8;
      }
      if (p.vout && l31 < p.d) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tk = ktok[wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
          if (tk >= 0) p.vout[(long long)tk * p.ldv + p.v_off + g * p.d + l31] = acc[r];
        }
      }
      ++it;
    }
    WF_T(7 + 8 * hi);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // K, V^T and the bias table of this head are complete
    WF_T(8 + 8 * hi);

    // ---- attention: 8 chunks of 32 keys, online softmax (one S^T tile live at a time: the x fragments own 96 registers) --
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    // a rolled loop: one tile's bias / K / V operands in flight at a time (the partner wave on the SIMD covers the latencies);
    // per tile the table pointer moves up by (32 / WW) key rows, the offsets inside a tile are immediates
    const float* relp = rel_base + CMAX;
    const unsigned char* kp = KB + l31 * WF_KROWB + 16 * hh;
    const unsigned char* vq = VB + l31 * WF_VROWB + 16 * hh;
    const unsigned* samep = same + 8 * qreg_id;
#pragma unroll UNR
    for (int t = 0; t < 8; ++t, relp -= (32 / WW) * RS, kp += WF_KWAVE, vq += 64) {
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2);                          // key inside the tile (lane half adds 4 to kx)
        st[r] = relp[-((k / WW) * RS + (k % WW))];
      }
      // this tile's V^T operands are requested now and consumed after the softmax: their LDS latency hides behind QK^T
      bf16x8 vh2[2], vl2[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        vh2[s] = *reinterpret_cast<const bf16x8*>(vq + 32 * s);
        if (NTERMS == 3) vl2[s] = *reinterpret_cast<const bf16x8*>(vq + 32 * s + WF_VPLB);
      }
      {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(kp + 32 * s);
          if (NTERMS == 3) {
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(kp + 32 * s + 32 * WF_KROWB);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[s], st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[s], st, 0, 0, 0);
          }
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[s], st, 0, 0, 0);
        }
      }
      if (blk_mask) {                                   // keys outside the query's shift region get -100 (hat_arch.py:921-940)
        const unsigned m = samep[t] >> (4 * hh);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          st[r] += ((m >> ((r & 3) + 8 * (r >> 2))) & 1u) ? 0.f : -100.0f;
      }
      float mx = st[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float corr = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
      const float mneg = -m_new * LOG2E;
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= corr;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 ph, pl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[8 * s + j], LOG2E, mneg));
          ls += e;
          const __bf16 h = (__bf16)e;
          ph[j] = h;
          if (NTERMS == 3) pl[j] = (__bf16)(e - (float)h);
        }
        if (NTERMS == 3) {
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh2[s], pl, o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl2[s], ph, o, 0, 0, 0);
        }
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh2[s], ph, o, 0, 0, 0);
      }
      l_run = l_run * corr + ls;
      m_run = m_new;
    }
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));
    WF_T(9 + 8 * hi);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave is done with K / V^T / the bias table of this head
    WF_T(10 + 8 * hi);

    // ---- O^T -> rows: transpose through the wave's own K rows (private to it until its next k tile) -----------------
    {
      float* tr = reinterpret_cast<float*>(KB + wid * WF_KWAVE);       // [32 queries][34]
#pragma unroll
      for (int r = 0; r < 16; ++r) tr[l31 * 34 + (r & 3) + 8 * (r >> 2) + 4 * hh] = o[r] * inv;
      const int pr = lane & 15;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = (lane >> 4) + 4 * i;
        const int tk = ktok[wid * 32 + q];
        const float2 v2 = *reinterpret_cast<const float2*>(tr + q * 34 + 2 * pr);
        if (tk >= 0 && 2 * pr < p.d)
          *reinterpret_cast<float2*>(p.out + (long long)tk * p.ldo + p.o_off + g * p.d + 2 * pr) = v2;
      }
    }
  }
  WF_T(60);
#ifdef WF_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WF_T(61);
#endif
}

#include "win_attn_fused_v2.inc"

extern "C" int ff_win_attn_fused(const float* x, int ldx, float* out, int ldo, int o_off, const float* gamma, const float* beta,
                                 float eps, const void* w_tiles, const float* bias_padded, const float* rel_padded, int rel_rows,
                                 int rel_stride, int B, int H, int W, int Hp, int Wp, int wh, int ww, int shift_h, int shift_w,
                                 int use_mask, int head0, int nheads, int d, int K, int zero_pad_tokens, float* xn_out, int ldxn,
                                 float* v_out, int ldv, int v_off, int nterms, int out_bf16, int xn_bf16, void* stream) {
  FF_CHECK_ARG(x && out && w_tiles && rel_padded, "ff_win_attn_fused: null pointer");
  FF_CHECK_ARG(nterms == 1 || nterms == 3, "ff_win_attn_fused: nterms must be 1 or 3");
  FF_CHECK_ARG((!out_bf16 && !xn_bf16) || nterms == 1, "ff_win_attn_fused: bf16 outputs exist for nterms == 1 only");
  FF_CHECK_ARG(wh * ww == 256 && (ww == 8 || ww == 16 || ww == 32), "ff_win_attn_fused: window must hold 256 tokens, width 8 / 16 / 32 (got %dx%d)", wh, ww);
  FF_CHECK_ARG(K > 0 && K <= 192 && K % 4 == 0 && ldx >= K && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0, "ff_win_attn_fused: x rows must be <= 192 wide and 16-byte aligned");
  FF_CHECK_ARG(d > 0 && d <= 32 && d % 2 == 0 && nheads > 0 && nheads <= 6 && head0 >= 0, "ff_win_attn_fused: head dim %d / heads %d unsupported", d, nheads);
  FF_CHECK_ARG(Hp % wh == 0 && Wp % ww == 0 && Hp >= H && Wp >= W, "ff_win_attn_fused: padded dims must tile by the window");
  FF_CHECK_ARG(shift_h >= 0 && shift_w >= 0 && shift_h < wh && shift_w < ww, "ff_win_attn_fused: bad shift");
  FF_CHECK_ARG(!use_mask || (shift_h > 0 && shift_w > 0), "ff_win_attn_fused: mask needs a shift");
  FF_CHECK_ARG(rel_rows == 2 * wh - 1 && rel_stride == (ww == 8 ? 40 : (ww == 16 ? 48 : 64)) && rel_rows * rel_stride <= WF_REL_MAX, "ff_win_attn_fused: bias table must be [heads][2wh-1][stride 40 / 48 / 64 for window width 8 / 16 / 32]");
  FF_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "ff_win_attn_fused: gamma/beta come together");
  FF_CHECK_ARG((((uintptr_t)w_tiles) & 15) == 0 && (!gamma || ((((uintptr_t)gamma) & 15) == 0 && (((uintptr_t)beta) & 15) == 0)), "ff_win_attn_fused: weights / gamma / beta must be 16-byte aligned");
  FF_CHECK_ARG(ldo % 2 == 0 && o_off % 2 == 0 && (((uintptr_t)out) & 7) == 0 && ldo >= o_off + (head0 + nheads) * d, "ff_win_attn_fused: out rows must be 8-byte aligned and hold every head");
  FF_CHECK_ARG(!xn_out || (gamma && ldxn >= K && ldxn % 4 == 0 && (((uintptr_t)xn_out) & 15) == 0), "ff_win_attn_fused: xn_out needs LayerNorm parameters and 16-byte aligned rows");
  FF_CHECK_ARG(!v_out || ldv >= v_off + (head0 + nheads) * d, "ff_win_attn_fused: v_out rows too short");
  FF_CHECK_ARG((long long)B * H * W < (1LL << 31), "ff_win_attn_fused: too many tokens");
  WinFusedParams p;
  p.x = x; p.out = out; p.gamma = gamma; p.beta = beta; p.w = (const __bf16*)w_tiles; p.bias = bias_padded; p.rel = rel_padded;
  p.xn = xn_out; p.vout = v_out; p.ldx = ldx; p.ldo = ldo; p.ldxn = ldxn; p.ldv = ldv; p.o_off = o_off; p.v_off = v_off;
  p.B = B; p.H = H; p.W = W; p.Hp = Hp; p.Wp = Wp; p.sh = shift_h; p.sw = shift_w; p.use_mask = use_mask;
  p.nwx = Wp / ww; p.nwy = Hp / wh; p.head0 = head0; p.nheads = nheads; p.d = d; p.K = K; p.zero_pad = zero_pad_tokens;
  p.rel_rows = rel_rows; p.rel_stride = rel_stride; p.eps = eps; p.out_bf16 = out_bf16; p.xn_bf16 = xn_bf16;
#ifdef WF_TIMING
  p.dbg = g_wf_dbg;
#endif
  const long long nblk = (long long)B * p.nwx * p.nwy;
  FF_CHECK_ARG(nblk < (1LL << 31), "ff_win_attn_fused: grid too large");
  static_assert(WF_LDS <= 160 * 1024, "LDS image too large");
  static_assert(8 * 32 * WF_XS_ROW * 4 <= 8 * WF_KWAVE + 2 * WF_VPLB, "gather patch must fit in the K / V area");
  static int unr = -1, prio = -1, v2 = -1;
  if (v2 < 0) { const char* e = getenv("FF_WF_V2"); v2 = (e && e[0] == '0') ? 0 : 1; }
  if (unr < 0) { const char* e = getenv("FF_WF_UNROLL"); unr = (e && e[0] == '1') ? 1 : 2; const char* q = getenv("FF_WF_PRIO"); prio = (q && q[0] == '1') ? 1 : 0; }
  p.prio = prio;
#define WF_LAUNCH1(WWV, NT, U)                                                                                                \
  do {                                                                                                                        \
    static bool attr_set = false;                                                                                             \
    if (!attr_set) {                                                                                                          \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&win_attn_fused_kernel<WWV, NT, U>),                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, WF_LDS);                                 \
      if (e != hipSuccess) { ff_set_error("ff_win_attn_fused: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; } \
      attr_set = true;                                                                                                        \
    }                                                                                                                         \
    hipLaunchKernelGGL((win_attn_fused_kernel<WWV, NT, U>), dim3((unsigned)nblk), dim3(512), WF_LDS, (hipStream_t)stream, p); \
  } while (0)
#define WF_LAUNCH(WWV, NT) do { if (unr == 1) WF_LAUNCH1(WWV, NT, 1); else WF_LAUNCH1(WWV, NT, 2); } while (0)
  static_assert(W2_LDS <= 160 * 1024, "LDS image too large");
  static_assert(8 * 32 * WF_XS_ROW * 4 <= 2 * W2_KBUF + 2 * W2_VBUF, "gather patch must fit in the K / V area");
#define W2_LAUNCH1(WWV, ON)                                                                                                   \
  do {                                                                                                                        \
    static bool attr_set = false;                                                                                             \
    if (!attr_set) {                                                                                                          \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&win_attn_fused_v2_kernel<WWV, ON>),                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS);                                 \
      if (e != hipSuccess) { ff_set_error("ff_win_attn_fused: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; } \
      attr_set = true;                                                                                                        \
    }                                                                                                                         \
    hipLaunchKernelGGL((win_attn_fused_v2_kernel<WWV, ON>), dim3((unsigned)nblk), dim3(512), W2_LDS, (hipStream_t)stream, p); \
  } while (0)
#define W2_LAUNCH(WWV) do { if (d == 30) W2_LAUNCH1(WWV, true); else W2_LAUNCH1(WWV, false); } while (0)
  FF_CHECK_ARG((!out_bf16 && !xn_bf16) || v2, "ff_win_attn_fused: bf16 outputs need the v2 kernel (FF_WF_V2)");
  FF_CHECK_ARG(!xn_bf16 || (ldxn % 8 == 0), "ff_win_attn_fused: bf16 xn rows must be 16-byte aligned");
  if (nterms == 1 && v2) {
    FF_CHECK_ARG((((uintptr_t)rel_padded) & 15) == 0, "ff_win_attn_fused: the bias table must be 16-byte aligned");
    if (ww == 8) W2_LAUNCH(8); else if (ww == 16) W2_LAUNCH(16); else W2_LAUNCH(32);
  } else
  if (nterms == 3) { if (ww == 8) WF_LAUNCH(8, 3); else if (ww == 16) WF_LAUNCH(16, 3); else WF_LAUNCH(32, 3); }
  else { if (ww == 8) WF_LAUNCH(8, 1); else if (ww == 16) WF_LAUNCH(16, 1); else WF_LAUNCH(32, 1); }
#undef WF_LAUNCH1
#undef WF_LAUNCH
#undef W2_LAUNCH1
#undef W2_LAUNCH
  FF_LAUNCH_CHECK("ff_win_attn_fused");
  return FF_OK;
}

// HAT overlapping cross-attention (OCAB), attention stage, plain bf16:  16x16 query windows against 24x24 key windows
//   hat_arch.py:392-438 (OCAB.forward: unfold of k / v with kernel 24, stride 16, padding 4 -- zero keys outside the image --,
//   relative_position_bias_table[rpi_oca], softmax, attn @ v) -- the qkv projection before and the proj + MLP after are separate launches.
// One persistent workgroup per WINDOW walks the heads; what the two-stage kernel (attention_bf16.hip, one workgroup per window and
// head) paid for and this one does not:
//   * the bias came from an expanded [heads][576][256] table, 590 KB re-read from L2 per workgroup (3.3x the launch's algorithmic
//     traffic).  Here the head's compact 39 x 39 table sits in LDS, and the keys are walked in 4-row x 8-column tiles of the
//     24 x 24 window, so a lane's sixteen scores of a tile read at compile-time offsets from one base (row r>>2, column r&3 of the
//     tile): eight ds_read2_b32, no per-key offset table;
//   * K / V staging and attention ran back to back between two barriers per 128 keys.  Here a head's keys are two stages of nine
//     32-key tiles, double buffered: the rows of the next stage (or of the next head's first stage) are in flight in registers while
//     the current stage is computed -- one barrier per 288 keys;
//   * softmax as in win_attn_fused_v2: denominator through the padded V channel 30, running maximum exchanged with
//     v_permlane32_swap, rescale skipped while no maximum moves, output stored straight from the accumulators.
// The reference indexes its bias table with NEGATIVE numbers (offset a = ws - ows + 1 = -7) that PyTorch wraps; the table arrives
// rotated (prep.pack_rel_overlap) so that  i = 600 - 39 qy - qx + 39 ky + kx  is in [0, 1521).
#include "ff_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define OC_NK 576
#define OC_RELW 39
#define OC_RELN 1521
#define OC_STK 288                        // keys per stage: nine 32-key tiles = rows 12 s .. 12 s + 11 of the key window
#define OC_KROWB 80                       // K row: 32 bf16 + 16 B pad
#define OC_VROWB 592                      // V^T row: 288 bf16 + 16 B pad (148 dwords: conflict-free ds_read_b128 across channels)
#define OC_KBUF (OC_STK * OC_KROWB)
#define OC_VBUF (32 * OC_VROWB)
#define OC_STAGEB (OC_KBUF + OC_VBUF)
#define OC_OFF_REL (2 * OC_STAGEB)
#define OC_RELB (1536 * 4)
#define OC_OFF_TOK (OC_OFF_REL + 2 * OC_RELB)
#define OC_LDS (OC_OFF_TOK + OC_NK * 4)
#define OC_NLOAD 18                       // 288 keys / 16 keys per pass

struct OcabParams {
  const float* qkv; float* out; const float* rel;
  int ldq, ldo, q_off, k_off, v_off, o_off, B, H, W, nwx, nwy, heads, out_bf16;
  float scale;
#ifdef OC_TIMING
  unsigned long long* dbg;   // tools/oc_time.cpp: [block][wave][64] wall-clock stamps (debug build only)
#endif
};
#ifdef OC_TIMING
static unsigned long long* g_oc_dbg = nullptr;
#define OC_T(i) do { if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[((long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 64 + (i)] = wall_clock64(); } while (0)
#else
#define OC_T(i) do { } while (0)
#endif

typedef __attribute__((address_space(3))) const float* oc_lds_cf;

// the value held by lanes 0..31 / 32..63, in every lane (see win_attn_fused_v2.inc: inline assembly because the builtin of this
// toolchain returns its first result twice)
__device__ __forceinline__ void oc_halves(float v, float& lo, float& hi) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  lo = a; hi = b;
}
__device__ __forceinline__ int oc_swap23(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }

__global__ __launch_bounds__(512) void ocab_attn_kernel(OcabParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* ktok = reinterpret_cast<int*>(smem + OC_OFF_TOK);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  int bid = blockIdx.x;
  const int wx = bid % p.nwx; bid /= p.nwx;
  const int wy = bid % p.nwy;
  const int b = bid / p.nwy;
  OC_T(0);

  // ---- key table in TILE order: key index 32 T + kappa, T = 3 tr + tc, is window position (4 tr + kappa / 8, 8 tc + kappa % 8) ---
  for (int kidx = tid; kidx < OC_NK; kidx += 512) {
    const int T = kidx >> 5, kap = kidx & 31;
    const int ky = 4 * (T / 3) + (kap >> 3), kx = 8 * (T % 3) + (kap & 7);
    const int y = wy * 16 - 4 + ky, x = wx * 16 - 4 + kx;
    ktok[kidx] = ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) ? (b * p.H + y) * p.W + x : -1;
  }
  // constant parts of both stage buffers: K channels 30 / 31 = 0, V^T row 30 = 1 (the denominator channel), row 31 = 0
  for (int i = tid; i < 2 * OC_STK; i += 512)
    *reinterpret_cast<unsigned*>(smem + (i / OC_STK) * OC_STAGEB + (i % OC_STK) * OC_KROWB + 60) = 0u;
  for (int i = tid; i < 2 * 2 * (OC_VROWB / 4); i += 512) {
    const int buf = i / (2 * (OC_VROWB / 4)), r = i % (2 * (OC_VROWB / 4));
    const int row = 30 + r / (OC_VROWB / 4), col = r % (OC_VROWB / 4);
    *reinterpret_cast<unsigned*>(smem + buf * OC_STAGEB + OC_KBUF + row * OC_VROWB + 4 * col) = row == 30 ? 0x3F803F80u : 0u;
  }
  __syncthreads();

  const int qi = wid * 32 + l31;
  const int qy = qi >> 4, qx = qi & 15;
  const int ty = wy * 16 + qy, tx = wx * 16 + qx;
  const bool qvalid = ty < p.H && tx < p.W;
  const long long qtok = qvalid ? (long long)(b * p.H + ty) * p.W + tx : 0;
  const int rel_lane = 600 - OC_RELW * qy - qx + 4 * hh;
  const float LOG2E = 1.4426950408889634f;

  // ---- staging: lane & 31 = float2 pair of a key's head slice (0..14 k, 15..29 v, 30 / 31 idle), (tid >> 5) + 16 i = key of the stage:
  //      no per-element index arithmetic, a wave instruction fetches the four 120-byte k / v segments of two keys --------------------
  const int spr = tid & 31, skey = tid >> 5;
  const bool s_on = spr < 30, s_isk = spr < 15;
  const int s_ch = s_isk ? p.k_off + 2 * spr : p.v_off + 2 * (spr - 15);
  float2 stg[OC_NLOAD];
  float tb[3];
  float2 qn[8];
  unsigned stg_ok = 0u;                                 // validity of the staged keys (bit i): applied when the stage is STORED -- a select
                                                        // on the loaded value here would make every load wait for its data (measured: 3.5 us
                                                        // per stage just to "issue" the loads, tools/oc_time.cpp)
  // in two halves: a wave stalls at issue once it has more than a dozen or so loads outstanding (tools/oc_time.cpp: "issuing" all 18 took
  // 3.6 us = the latency of the first ones); the second half is issued a third of the way into the stage's tiles
  auto stage_load = [&](int h, int st, int part) {
    if (part == 0) stg_ok = 0u;
#pragma unroll
    for (int i = 0; i < OC_NLOAD; ++i) {
      if ((i < OC_NLOAD / 2) != (part == 0)) continue;
      const int tk = s_on ? ktok[st * OC_STK + skey + 16 * i] : -1;
      stg_ok |= (tk >= 0 ? 1u : 0u) << i;
      stg[i] = *reinterpret_cast<const float2*>(p.qkv + (tk >= 0 ? (long long)tk * p.ldq + s_ch + h * 30 : 0));
    }
  };
  auto stage_store = [&](int buf) {
    unsigned char* kb = smem + buf * OC_STAGEB;
    unsigned char* vb = kb + OC_KBUF;
    if (s_on) {
#pragma unroll
      for (int i = 0; i < OC_NLOAD; ++i) {
        const int kl = skey + 16 * i;
        const bool ok = (stg_ok >> i) & 1u;
        const __bf16 a = (__bf16)(ok ? stg[i].x : 0.f), c = (__bf16)(ok ? stg[i].y : 0.f);
        if (s_isk) {
          const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, c) << 16);
          *reinterpret_cast<unsigned*>(kb + kl * OC_KROWB + 4 * spr) = pk;
        } else {
          const int ch = 2 * (spr - 15), pos = oc_swap23(kl);
          *reinterpret_cast<__bf16*>(vb + ch * OC_VROWB + 2 * pos) = a;
          *reinterpret_cast<__bf16*>(vb + (ch + 1) * OC_VROWB + 2 * pos) = c;
        }
      }
    }
  };
  auto head_load = [&](int h) {                        // bias table and queries of head h
#pragma unroll
    for (int i = 0; i < 3; ++i) tb[i] = (tid + 512 * i < OC_RELN) ? p.rel[(long long)h * OC_RELN + tid + 512 * i] : 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dd = 16 * s + 8 * hh + 2 * j;
        qn[4 * s + j] = dd < 30 ? *reinterpret_cast<const float2*>(p.qkv + qtok * p.ldq + p.q_off + h * 30 + dd) : (float2){0.f, 0.f};
      }
  };
  bf16x8 qh[2];
  auto head_store = [&](int h) {
    float* tl = reinterpret_cast<float*>(smem + OC_OFF_REL + (h & 1) * OC_RELB);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (tid + 512 * i < OC_RELN) tl[tid + 512 * i] = tb[i];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = qvalid;
        qh[s][2 * j] = (__bf16)(ok ? qn[4 * s + j].x * p.scale : 0.f);
        qh[s][2 * j + 1] = (__bf16)(ok ? qn[4 * s + j].y * p.scale : 0.f);
      }
  };

  stage_load(0, 0, 0);
  stage_load(0, 0, 1);
  head_load(0);
  stage_store(0);
  head_store(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  OC_T(1);
  f32x16 o;
  float m_run = -INFINITY;
  const int nsteps = 2 * p.heads;
  for (int n = 0; n < nsteps; ++n) {
    const int h = n >> 1, st = n & 1, buf = n & 1;
    const bool more = n + 1 < nsteps;
    if (more) {
      stage_load((n + 1) >> 1, (n + 1) & 1, 0);
      if (st == 1) head_load(h + 1);
    }
    if (n < 4) OC_T(2 + 4 * n);
    if (st == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      m_run = -INFINITY;
    }
    const unsigned char* kb = smem + buf * OC_STAGEB + l31 * OC_KROWB + 16 * hh;
    const unsigned char* vb = smem + buf * OC_STAGEB + OC_KBUF + l31 * OC_VROWB + 16 * hh;
    const unsigned rel0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(smem + OC_OFF_REL + (h & 1) * OC_RELB) +
                          4u * (unsigned)(rel_lane + st * (12 * OC_RELW));
    for (int tq = 0; tq < 3; ++tq) {
      if (more && tq == 1) stage_load((n + 1) >> 1, (n + 1) & 1, 1);
#pragma unroll
      for (int tm = 0; tm < 3; ++tm) {
        const int t = 3 * tq + tm;
        unsigned ra = rel0 + 4u * (unsigned)(tq * 4 * OC_RELW + tm * 8);
        asm volatile("" : "+v"(ra));                    // one base register per tile: the gathers below use immediate offsets
        const oc_lds_cf relp = (oc_lds_cf)(size_t)ra;
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = relp[(r >> 2) * OC_RELW + (r & 3)];
        bf16x8 vh2[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) vh2[s] = *reinterpret_cast<const bf16x8*>(vb + t * 64 + 32 * s);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(kb + t * 32 * OC_KROWB + 32 * s);
          sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[s], sc, 0, 0, 0);
        }
        float mx = sc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
        {
          float lo, hi;
          oc_halves(mx, lo, hi);
          mx = fmaxf(lo, hi);
        }
        if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0) {
          const float m_new = fmaxf(m_run, mx);
          const float corr = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
#pragma unroll
          for (int r = 0; r < 16; ++r) o[r] *= corr;
          m_run = m_new;
        }
        const float mneg = -m_run * LOG2E;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 ph;
#pragma unroll
          for (int j = 0; j < 8; ++j) ph[j] = (__bf16)__builtin_amdgcn_exp2f(__builtin_fmaf(sc[8 * s + j], LOG2E, mneg));
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh2[s], ph, o, 0, 0, 0);
        }
      }
    }
    if (n < 4) OC_T(3 + 4 * n);
    if (st == 1) {                                       // head finished: channel 30 = register 14 of the upper half-wave holds sum(P)
      float lo, den;
      oc_halves(o[14], lo, den);
      const float inv = 1.0f / den;
      if (qvalid && p.out_bf16) {                       // bf16 rows (ldo in elements): the consumer rounds to bf16 anyway
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        __bf16* orow = reinterpret_cast<__bf16*>(p.out) + qtok * p.ldo + p.o_off + h * 30 + 4 * hh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = 8 * i + 4 * hh;
          if (c < 30) *reinterpret_cast<bf16x2*>(orow + 8 * i) = (bf16x2){(__bf16)(o[4 * i] * inv), (__bf16)(o[4 * i + 1] * inv)};
          if (c + 2 < 30) *reinterpret_cast<bf16x2*>(orow + 8 * i + 2) = (bf16x2){(__bf16)(o[4 * i + 2] * inv), (__bf16)(o[4 * i + 3] * inv)};
        }
      } else if (qvalid) {
        float* orow = p.out + qtok * p.ldo + p.o_off + h * 30 + 4 * hh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = 8 * i + 4 * hh;
          if (c < 30) *reinterpret_cast<float2*>(orow + 8 * i) = (float2){o[4 * i] * inv, o[4 * i + 1] * inv};
          if (c + 2 < 30) *reinterpret_cast<float2*>(orow + 8 * i + 2) = (float2){o[4 * i + 2] * inv, o[4 * i + 3] * inv};
        }
      }
    }
    if (more) {
      stage_store(buf ^ 1);
      if (st == 1) head_store(h + 1);
    }
    if (n < 4) OC_T(4 + 4 * n);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // the next stage is in LDS; every wave is done with this one
    if (n < 4) OC_T(5 + 4 * n);
  }
  OC_T(60);
}

extern "C" int ff_ocab_attn(const float* qkv, int ldq, int q_off, int k_off, int v_off, float* out, int ldo, int o_off,
                            const float* rel_rotated, int B, int H, int W, int heads, int d, int ws, int ows, float scale, int out_bf16, void* stream) {
  FF_CHECK_ARG(qkv && out && rel_rotated, "ff_ocab_attn: null pointer");
  FF_CHECK_ARG(ws == 16 && ows == 24 && d == 30 && heads > 0, "ff_ocab_attn: built for 16x16 query / 24x24 key windows and head dim 30 (got %d / %d / %d)", ws, ows, d);
  FF_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0, "ff_ocab_attn: H and W must be multiples of the window");
  FF_CHECK_ARG(ldq % 2 == 0 && q_off % 2 == 0 && k_off % 2 == 0 && v_off % 2 == 0 && (((uintptr_t)qkv) & 7) == 0, "ff_ocab_attn: qkv rows must be 8-byte aligned");
  FF_CHECK_ARG(ldo % 2 == 0 && o_off % 2 == 0 && (((uintptr_t)out) & 7) == 0 && ldo >= o_off + heads * d, "ff_ocab_attn: out rows must be 8-byte aligned and hold every head");
  FF_CHECK_ARG((long long)B * H * W * (ldq > ldo ? ldq : ldo) < (1LL << 31), "ff_ocab_attn: tensor too large for 32-bit token offsets");
  OcabParams p;
  p.qkv = qkv; p.out = out; p.rel = rel_rotated; p.ldq = ldq; p.ldo = ldo; p.q_off = q_off; p.k_off = k_off; p.v_off = v_off; p.o_off = o_off;
  p.B = B; p.H = H; p.W = W; p.nwx = W / 16; p.nwy = H / 16; p.heads = heads; p.scale = scale; p.out_bf16 = out_bf16;
#ifdef OC_TIMING
  p.dbg = g_oc_dbg;
#endif
  const long long nblk = (long long)B * p.nwx * p.nwy;
  FF_CHECK_ARG(nblk < (1LL << 31), "ff_ocab_attn: grid too large");
  static_assert(OC_LDS <= 160 * 1024, "LDS image too large");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ocab_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, OC_LDS);
    if (e != hipSuccess) { ff_set_error("ff_ocab_attn: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; }
    attr_set = true;
  }
  hipLaunchKernelGGL(ocab_attn_kernel, dim3((unsigned)nblk), dim3(512), OC_LDS, (hipStream_t)stream, p);
  FF_LAUNCH_CHECK("ff_ocab_attn");
  return FF_OK;
}

// Window attention on the bf16 matrix cores with split operands (same contract as attention.hip /
// ff_window_attn; selected by ff_window_attn_bf16s).  v_mfma_f32_32x32x16_bf16, fp32 accumulate:
//   S^T = K Q^T      K tile row-major [key][d] in LDS (80-byte rows: conflict-free ds_read_b128), Q in registers
//   O^T += V^T P^T   V staged TRANSPOSED [d][key] with the key order inside each 16-key group permuted
//                    (bits 2 and 3 swapped) so that the 8 keys a lane needs for one k-step are one ds_read_b128
//                    and line up with the accumulator registers 8s..8s+7 that hold P^T (guide section 3,
//                    'An accumulator tile as the next MFMA's operand': element j of lane half h is key
//                    16s + 8(j>>2) + 4h + (j&3)).
// NTERMS = 3: every product is hi*hi + lo*hi + hi*lo of bf16 splits (fp32-grade); NTERMS = 1: plain bf16.
// Softmax in base 2: p = exp2(fma(s, log2e, -m*log2e)) -- one v_fma + one v_exp_f32 per element.
#include "ff_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct AttnBfParams {
  const float* qkv;
  float* out;
  const float* biasT;
  const float* rel;      // optional compact relative-position table [heads][(2wh-1)*(2ww-1)] (NULL: use biasT)
  int rel_w, rel_n, lkw;  // ww+kw-1, table entries per head, log2(kw)
  int rel_gen, rel_base;  // overlapping keys (kh != wh): per-key offset table in LDS + rotated table (see the launcher)
  int ldq, ldo;
  int q_off, k_off, v_off, o_off;
  int B, H, W, Hp, Wp;
  int wh, ww, kh, kw;
  int sh, sw;
  int use_mask;
  int nkpad;
  int heads, d;
  float scale;
  int nwx, nwy;
  int vec_out;     // float2 output rows: d, o_off, ldo even and out 8-byte aligned
  int vec_q;       // float2 query loads: d, q_off, ldq even and qkv 8-byte aligned
};

#define AKC 128            // keys per chunk
#define KROWB 80           // K row: 32 bf16 + 16 B pad
#define VROWB 272          // V^T row: 128 bf16 + 16 B pad

__device__ __forceinline__ int swap23(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }

template <int NTERMS>
__global__ __launch_bounds__(512) void window_attn_bf16_kernel(AttnBfParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Kh = smem;                          // [AKC][KROWB]
  unsigned char* Kl = Kh + AKC * KROWB;
  unsigned char* Vh = Kl + AKC * KROWB;              // [32][VROWB]
  unsigned char* Vl = Vh + 32 * VROWB;
  int* ktok = reinterpret_cast<int*>(Vl + 32 * VROWB);   // [nchunks*AKC] token index of every key (or -1: zero key)
  int* kregAll = ktok + p.nkpad;                         // [nchunks*AKC] shift-region id of every key
  float* Tl = reinterpret_cast<float*>(kregAll + p.nkpad);   // [rel_n] this head's relative-position table (rel path)
  int* kofs = reinterpret_cast<int*>(Tl + (p.rel ? p.rel_n : 0));   // [nkpad] ky * rel_w + kx of every key (rel_gen path)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;

  int bid = blockIdx.x;
  const int head = bid % p.heads; bid /= p.heads;
  const int wx = bid % p.nwx; bid /= p.nwx;
  const int wy = bid % p.nwy;
  const int b = bid / p.nwy;
  const int nk = p.kh * p.kw;
  const int koy = wy * p.wh - (p.kh - p.wh) / 2, kox = wx * p.ww - (p.kw - p.ww) / 2;

  const int qi = wid * 32 + l31;
  const int qy = wy * p.wh + qi / p.ww, qx = wx * p.ww + qi % p.ww;
  int oy = qy + p.sh, ox = qx + p.sw;
  if (oy >= p.Hp) oy -= p.Hp;
  if (ox >= p.Wp) ox -= p.Wp;
  const bool qvalid = oy < p.H && ox < p.W;
  const long long qtok = ((long long)b * p.H + oy) * p.W + ox;
  int qreg_id = 0;
  if (p.use_mask) {
    const int ry = qy < p.Hp - p.wh ? 0 : (qy < p.Hp - p.sh ? 1 : 2);
    const int rx = qx < p.Wp - p.ww ? 0 : (qx < p.Wp - p.sw ? 1 : 2);
    qreg_id = 3 * ry + rx;
  }
  bf16x8 qh[2], ql[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float qf[8];
    if (p.vec_q) {                                  // 8-byte loads: the head's channels start on an even float
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const int dd = 16 * s + 8 * hh + j;
        const bool ok = qvalid && dd < p.d;
        const float2 u = *reinterpret_cast<const float2*>(p.qkv + (ok ? qtok * p.ldq + p.q_off + head * p.d + dd : 0));
        qf[j] = ok ? u.x * p.scale : 0.f;
        qf[j + 1] = ok ? u.y * p.scale : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int dd = 16 * s + 8 * hh + j;
        qf[j] = (qvalid && dd < p.d) ? p.qkv[qtok * p.ldq + p.q_off + head * p.d + dd] * p.scale : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 h = (__bf16)qf[j];
      qh[s][j] = h;
      ql[s][j] = (__bf16)(qf[j] - (float)h);
    }
  }

  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float LOG2E = 1.4426950408889634f;

  const int nchunks = (nk + AKC - 1) / AKC;
  // ---- per-block key tables: token index (or -1 for the zero keys outside the image) and mask region of every key ------
  for (int kidx = tid; kidx < nchunks * AKC; kidx += 512) {
    int tk = -1, rid = 0;
    if (kidx < nk) {
      const int ky = koy + kidx / p.kw, kx = kox + kidx % p.kw;
      if (ky >= 0 && kx >= 0 && ky < p.Hp && kx < p.Wp) {
        int yy = ky + p.sh, xx = kx + p.sw;
        if (yy >= p.Hp) yy -= p.Hp;
        if (xx >= p.Wp) xx -= p.Wp;
        if (yy < p.H && xx < p.W) tk = (b * p.H + yy) * p.W + xx;
        if (p.use_mask) {
          const int ry = ky < p.Hp - p.wh ? 0 : (ky < p.Hp - p.sh ? 1 : 2);
          const int rx = kx < p.Wp - p.ww ? 0 : (kx < p.Wp - p.sw ? 1 : 2);
          rid = 3 * ry + rx;
        }
      }
    }
    ktok[kidx] = tk;
    kregAll[kidx] = rid;
    if (p.rel_gen) kofs[kidx] = kidx < nk ? (kidx / p.kw) * p.rel_w + kidx % p.kw : 0;
  }
  if (p.rel)
    for (int i = tid; i < p.rel_n; i += 512) Tl[i] = p.rel[(long long)head * p.rel_n + i];
  __syncthreads();
  // only windows in the last window row / column contain more than one shift region (mask is all zero elsewhere)
  const bool blk_mask = p.use_mask && (wy == p.nwy - 1 || wx == p.nwx - 1);
  // K/V rows of chunk c+1 are fetched while chunk c is computed; the bias of chunk c is fetched straight into the
  // S^T accumulators before the staging barrier -- one exposed L2 latency per chunk instead of three.
  float kv[8], vv[8];
  const long long hoff = head * p.d + l31;
  auto fetch_kv = [&](int c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int tk = ktok[c * AKC + wid * 2 + hh + 16 * i];
      const bool ok = tk >= 0 && l31 < p.d;
      const float* src = p.qkv + (long long)(ok ? tk : 0) * p.ldq + hoff;
      const float a = src[p.k_off], bq = src[p.v_off];
      kv[i] = ok ? a : 0.f;
      vv[i] = ok ? bq : 0.f;
    }
  };
  fetch_kv(0);
  for (int c = 0; c < nchunks; ++c) {
    const bool full = (c + 1) * AKC <= nk;            // every key of this chunk exists: no per-element tail handling
    // ---- bias of this chunk -> accumulators (C-in of the QK^T MFMAs) ---------------------------------------------
    f32x16 st[4];
    {
      // biasQ is quad-interleaved [heads][nk/4][256 queries][4 keys]: accumulator registers 4g..4g+3 of a lane are four
      // consecutive keys of its query, i.e. ONE 16-byte load (16 loads per chunk and lane instead of 64; a half-wave
      // reads 512 contiguous bytes).
      if (p.rel_gen) {
        // Overlapping key window (HAT OCAB, hat_arch.py:385-437: 16x16 queries, 24x24 keys): bias[q][k] = T[((ky - qy + a) * rel_w +
        // (kx - qx + a)) mod rel_n] with a = wh - kh + 1 < 0 -- the reference indexes its table with negative numbers and PyTorch
        // wraps them -- served from the ROTATED table U[i] = T[(i - m) mod rel_n] (prep.pack_rel_overlap), i = kofs[key] + lane part,
        // where kofs = ky * rel_w + kx sits in LDS (kw is not a power of two) and four consecutive keys are one 16-byte read.
        const int lane_base = p.rel_base - (qi / p.ww) * p.rel_w - (qi % p.ww);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int4 ko = *reinterpret_cast<const int4*>(kofs + c * AKC + t * 32 + 8 * g + 4 * hh);
            st[t][4 * g + 0] = Tl[lane_base + ko.x];
            st[t][4 * g + 1] = Tl[lane_base + ko.y];
            st[t][4 * g + 2] = Tl[lane_base + ko.z];
            st[t][4 * g + 3] = Tl[lane_base + ko.w];
          }
      } else if (p.rel) {
        // Relative-position bias gathered from the head's (2wh-1) x (2ww-1) table in LDS (3.8 KB) instead of streaming the
        // expanded [keys][queries] table (256 KB per head and window, 400 MB per launch from L2 / Infinity Cache):
        //   bias[q][k] = T[(qy - ky + wh - 1) * (2ww - 1) + (qx - kx + ww - 1)]        (hat_arch.py:882-899, dat_arch.py:300-318)
        // = lane part (query coordinates, key-column half 4 hh) + wave-uniform part per register (ky, kx of the key).
        const int lane_base = (qi / p.ww + p.wh - 1) * p.rel_w + (qi % p.ww) - 4 * hh + p.ww - 1;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key0 = c * AKC + t * 32 + (r & 3) + 8 * (r >> 2);        // uniform; lane half hh adds 4 to kx
            st[t][r] = Tl[lane_base - (key0 >> p.lkw) * p.rel_w - (key0 & (p.kw - 1))];
          }
      } else {
      const float* bbase = p.biasT + ((long long)head * nk + (long long)c * AKC) * 256;      // wave-uniform
      const int lo = hh * 1024 + 4 * qi;                                                      // floats: key half + query
      if (full) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(bbase + (t * 32 + 8 * g) * 256 + lo);
#pragma unroll
            for (int e = 0; e < 4; ++e) st[t][4 * g + e] = u[e];
          }
      } else {                           // ragged last chunk: clamp the key quad into the table (those keys are masked below)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            int k0 = c * AKC + t * 32 + 8 * g + 4 * hh;
            k0 = k0 < nk ? k0 : nk - 4;
            const f32x4 u = *reinterpret_cast<const f32x4*>(p.biasT + ((long long)head * nk + k0) * 256 + 4 * qi);
#pragma unroll
            for (int e = 0; e < 4; ++e) st[t][4 * g + e] = u[e];
          }
      }
      }
    }
    // ---- stage K (row-major) and V (transposed, bit-2/3 key permutation) of this chunk from the prefetched registers
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kk = wid * 2 + hh + 16 * i;
      const __bf16 k_h = (__bf16)kv[i], v_h = (__bf16)vv[i];
      *reinterpret_cast<__bf16*>(Kh + kk * KROWB + 2 * l31) = k_h;
      const int vp = swap23(kk);
      *reinterpret_cast<__bf16*>(Vh + l31 * VROWB + 2 * vp) = v_h;
      if (NTERMS == 3) {
        *reinterpret_cast<__bf16*>(Kl + kk * KROWB + 2 * l31) = (__bf16)(kv[i] - (float)k_h);
        *reinterpret_cast<__bf16*>(Vl + l31 * VROWB + 2 * vp) = (__bf16)(vv[i] - (float)v_h);
      }
    }
    __syncthreads();
    if (c + 1 < nchunks) fetch_kv(c + 1);

    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Kh + (t * 32 + l31) * KROWB + 32 * s + 16 * hh);
        if (NTERMS == 3) {
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(Kl + (t * 32 + l31) * KROWB + 32 * s + 16 * hh);
          st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[s], st[t], 0, 0, 0);
          st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[s], st[t], 0, 0, 0);
        }
        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[s], st[t], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    // ---- mask, running max ---------------------------------------------------------------------------------
    float mx = -INFINITY;
    if (blk_mask) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kregAll[c * AKC + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh] != qreg_id) st[t][r] += -100.0f;
    }
    if (!full) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (c * AKC + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= nk) st[t][r] = -INFINITY;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float corr = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
    const float mneg = -m_new * LOG2E;
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= corr;
    // ---- P = exp2(.), O^T += V^T P^T : accumulator registers 8s..8s+7 of a key tile are the B operand ------
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 ph, pl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][8 * s + j], LOG2E, mneg));
          ls += e;
          const __bf16 h = (__bf16)e;
          ph[j] = h;
          if (NTERMS == 3) pl[j] = (__bf16)(e - (float)h);
        }
        const bf16x8 vh = *reinterpret_cast<const bf16x8*>(Vh + l31 * VROWB + 2 * (t * 32 + 16 * s + 8 * hh));
        if (NTERMS == 3) {
          const bf16x8 vl = *reinterpret_cast<const bf16x8*>(Vl + l31 * VROWB + 2 * (t * 32 + 16 * s + 8 * hh));
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o, 0, 0, 0);
        }
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o, 0, 0, 0);
      }
    }
    l_run = l_run * corr + ls;
    m_run = m_new;
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (p.vec_out) {
    // O^T sits with the query in the lane and 16 head channels in registers: stored directly, every instruction would
    // scatter 64 dwords over 32 token rows.  The K/V staging buffers are idle after the loop's last barrier, so each wave
    // transposes its 32 x 32 tile through a private patch and writes 120-byte row segments as float2 (16 lanes per query).
    float* tr = reinterpret_cast<float*>(smem) + wid * (32 * 34);
    int* tokq = reinterpret_cast<int*>(smem + 8 * 32 * 34 * 4) + wid * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) tr[l31 * 34 + (r & 3) + 8 * (r >> 2) + 4 * hh] = o[r] * inv;
    if (hh == 0) tokq[l31] = qvalid ? (int)qtok : -1;
    const int pr = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = (lane >> 4) + 4 * i;
      const int tk = tokq[q];
      const float2 v = *reinterpret_cast<const float2*>(tr + q * 34 + 2 * pr);
      if (tk >= 0 && 2 * pr < p.d)
        *reinterpret_cast<float2*>(p.out + (long long)tk * p.ldo + p.o_off + head * p.d + 2 * pr) = v;
    }
    return;
  }
  if (qvalid) {
    float* op = p.out + qtok * p.ldo + p.o_off + head * p.d;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (dd < p.d) op[dd] = o[r] * inv;
    }
  }
}

extern "C" int ff_window_attn_bf16s(const float* qkv, int ldq, int q_off, int k_off, int v_off, float* out, int ldo,
                                    int o_off, const float* biasT, int B, int H, int W, int Hp, int Wp, int wh, int ww,
                                    int kh, int kw, int shift_h, int shift_w, int use_mask, int heads, int d, float scale,
                                    int nterms, const float* rel_table, void* stream) {
  FF_CHECK_ARG(qkv && out && (biasT || rel_table), "ff_window_attn_bf16s: null pointer");
  FF_CHECK_ARG(nterms == 1 || nterms == 3, "ff_window_attn_bf16s: nterms must be 1 or 3");
  FF_CHECK_ARG(rel_table || ((kh * kw) % 4 == 0 && (((uintptr_t)biasT) & 15) == 0), "ff_window_attn_bf16s: the quad-interleaved bias table needs kh*kw %% 4 == 0 and 16-byte alignment");
  const bool rel_gen = rel_table && (kh != wh || kw != ww);
  FF_CHECK_ARG(!rel_table || rel_gen || ((kw & (kw - 1)) == 0 && kw >= 8), "ff_window_attn_bf16s: the relative-position table needs a power-of-two window width >= 8 when keys == query window");
  FF_CHECK_ARG(!rel_gen || (kw % 8 == 0 && shift_h == 0 && shift_w == 0), "ff_window_attn_bf16s: the overlapping-window table needs kw %% 8 == 0 and no shift");
  FF_CHECK_ARG(wh * ww == 256, "ff_window_attn_bf16s: query window must hold 256 tokens (got %dx%d)", wh, ww);
  FF_CHECK_ARG(d > 0 && d <= 32 && heads > 0, "ff_window_attn_bf16s: head dim %d unsupported (<=32)", d);
  FF_CHECK_ARG(kh >= wh && kw >= ww && (kh - wh) % 2 == 0 && (kw - ww) % 2 == 0, "ff_window_attn_bf16s: bad key window");
  FF_CHECK_ARG(Hp % wh == 0 && Wp % ww == 0 && Hp >= H && Wp >= W, "ff_window_attn_bf16s: padded dims must tile by the window");
  FF_CHECK_ARG(shift_h >= 0 && shift_w >= 0 && shift_h < wh && shift_w < ww, "ff_window_attn_bf16s: bad shift");
  FF_CHECK_ARG(!(shift_h || shift_w) || (kh == wh && kw == ww), "ff_window_attn_bf16s: shift with overlapping keys unsupported");
  FF_CHECK_ARG(!use_mask || (shift_h > 0 && shift_w > 0), "ff_window_attn_bf16s: mask needs a shift");
  AttnBfParams p;
  p.qkv = qkv; p.out = out; p.biasT = biasT; p.ldq = ldq; p.ldo = ldo;
  // rel_table layouts: keys == query window: T [heads][(2wh-1)(2ww-1)] as stored by the reference.  Overlapping keys: the table
  // ROTATED by m = -(minimum index) = ((wh-1) - a)(rel_w + 1) with a = wh - kh + 1, so that the kernel's index
  //   kofs[key] + rel_base - qy rel_w - qx,   rel_base = m + a (rel_w + 1),   is always in [0, rel_n)  (prep.pack_rel_overlap).
  p.rel = rel_table; p.rel_w = ww + kw - 1; p.rel_n = (wh + kh - 1) * (ww + kw - 1); p.lkw = 0;
  p.rel_gen = rel_gen ? 1 : 0;
  p.rel_base = rel_gen ? ((wh - 1) - (wh - kh + 1)) * (p.rel_w + 1) + (wh - kh + 1) * (p.rel_w + 1) : 0;
  while ((1 << p.lkw) < kw) ++p.lkw;
  p.q_off = q_off; p.k_off = k_off; p.v_off = v_off; p.o_off = o_off;
  p.B = B; p.H = H; p.W = W; p.Hp = Hp; p.Wp = Wp; p.wh = wh; p.ww = ww; p.kh = kh; p.kw = kw;
  p.sh = shift_h; p.sw = shift_w; p.use_mask = use_mask; p.heads = heads; p.d = d; p.scale = scale;
  p.nwx = Wp / ww; p.nwy = Hp / wh;
  p.vec_q = (d % 2 == 0) && (q_off % 2 == 0) && (ldq % 2 == 0) && ((((uintptr_t)qkv) & 7) == 0);
  p.vec_out = (d % 2 == 0) && (o_off % 2 == 0) && (ldo % 2 == 0) && ((((uintptr_t)out) & 7) == 0);
  p.nkpad = (kh * kw + AKC - 1) / AKC * AKC;
  const long long nblk = (long long)B * p.nwx * p.nwy * heads;
  FF_CHECK_ARG(nblk < (1LL << 31), "ff_window_attn_bf16s: grid too large");
  FF_CHECK_ARG((long long)B * H * W < (1LL << 31), "ff_window_attn_bf16s: too many tokens");
  const size_t lds = (size_t)2 * AKC * KROWB + (size_t)2 * 32 * VROWB + (size_t)2 * p.nkpad * 4 + (rel_table ? (size_t)p.rel_n * 4 : 0) + (rel_gen ? (size_t)p.nkpad * 4 : 0);
  FF_CHECK_ARG(lds <= 64 * 1024, "ff_window_attn_bf16s: window too large for the LDS image");
  if (nterms == 3)
    hipLaunchKernelGGL(window_attn_bf16_kernel<3>, dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(window_attn_bf16_kernel<1>, dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, p);
  FF_LAUNCH_CHECK("ff_window_attn_bf16s");
  return FF_OK;
}

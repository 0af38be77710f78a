// Fused window attention on the matrix cores (fp32 MFMA), one workgroup per (window, head):
//   softmax( (q*scale) k^T + bias (+ shift mask) ) v
// covering HAT W-MSA / SW-MSA (hat_arch.py:165-196,266-303, mask :921-940), HAT OCAB
// (hat_arch.py:400-433: 16x16 queries against the 24x24 zero-padded overlapping key window) and the
// DAT rectangular-window branches (dat_arch.py:290-342,491-562: 8x32 / 32x8 windows on channel halves,
// zero tokens right/bottom of the image, optional cyclic shift + region mask).
// Window partition, cyclic shift, reverse shift, crop and the head split are all index arithmetic
// on the NHWC token tensor -- no roll / unfold / permute copies exist.
//
// Dataflow (per wave = 32 queries, 8 waves = the 256 queries of a window):
//   S^T[key][query] = K Q^T      A operand = K^T tile from LDS ([d][key], conflict free),
//                                B operand = Q (registers, pre-scaled)
//   accumulators hold S^T with the query on the lane and keys in registers, so
//   - the softmax row of a query is lane-local (+ one exchange with lane^32),
//   - each accumulator register is directly the B operand of O^T += V^T P^T (guide section 3,
//     'An accumulator tile as the next MFMA's operand'): P never touches LDS.
// Keys are processed in chunks of NKT*32 = 128 with an online softmax (2 chunks for 256 keys, 5 for OCAB's 576).
#include "ff_common.h"

struct AttnParams {
  const float* qkv;
  float* out;
  const float* biasT;  // [heads][nk][256]  (key-major, query fastest)
  int ldq, ldo;
  int q_off, k_off, v_off, o_off;
  int B, H, W, Hp, Wp;
  int wh, ww, kh, kw;
  int sh, sw;
  int use_mask;
  int heads, d;
  float scale;
  int nwx, nwy;
};

template <int NKT>
__global__ __launch_bounds__(512) void window_attn_kernel(AttnParams p) {
  constexpr int KC = NKT * 32;
  constexpr int KTS = KC + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Kt = smem;                         // [32][KTS]
  float* Vs = smem + 32 * KTS;              // [KC][32]
  int* kreg = reinterpret_cast<int*>(Vs + KC * 32);   // [KC] region id of each key (mask)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;

  int bid = blockIdx.x;
  const int head = bid % p.heads; bid /= p.heads;
  const int wx = bid % p.nwx; bid /= p.nwx;
  const int wy = bid % p.nwy;
  const int b = bid / p.nwy;

  const int nk = p.kh * p.kw;
  const int koy = wy * p.wh - (p.kh - p.wh) / 2, kox = wx * p.ww - (p.kw - p.ww) / 2;

  // ---- this lane's query token ------------------------------------------------------------
  const int qi = wid * 32 + l31;
  const int qy = wy * p.wh + qi / p.ww, qx = wx * p.ww + qi % p.ww;     // coords in the shifted grid
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
  float qv[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int dd = 2 * s + hh;
    qv[s] = (qvalid && dd < p.d) ? p.qkv[qtok * p.ldq + p.q_off + head * p.d + dd] * p.scale : 0.f;
  }

  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.biasT), 0, p.heads * nk * 256 * 4, 0x00020000);
  const int nchunks = (nk + KC - 1) / KC;
  for (int c = 0; c < nchunks; ++c) {
    // ---- stage K^T and V of this key chunk: 32 lanes = the d values of one key, 2 keys per wave pass
    for (int kk = wid * 2 + hh; kk < KC; kk += 16) {
      const int kidx = c * KC + kk;
      float kval = 0.f, vval = 0.f;
      int rid = 0;
      if (kidx < nk) {
        const int ky = koy + kidx / p.kw, kx = kox + kidx % p.kw;       // shifted-grid coords
        if (ky >= 0 && kx >= 0 && ky < p.Hp && kx < p.Wp) {
          int yy = ky + p.sh, xx = kx + p.sw;
          if (yy >= p.Hp) yy -= p.Hp;
          if (xx >= p.Wp) xx -= p.Wp;
          if (yy < p.H && xx < p.W && l31 < p.d) {
            const long long tok = ((long long)b * p.H + yy) * p.W + xx;
            kval = p.qkv[tok * p.ldq + p.k_off + head * p.d + l31];
            vval = p.qkv[tok * p.ldq + p.v_off + head * p.d + l31];
          }
          if (p.use_mask) {
            const int ry = ky < p.Hp - p.wh ? 0 : (ky < p.Hp - p.sh ? 1 : 2);
            const int rx = kx < p.Wp - p.ww ? 0 : (kx < p.Wp - p.sw ? 1 : 2);
            rid = 3 * ry + rx;
          }
        }
      }
      Kt[l31 * KTS + kk] = kval;
      Vs[kk * 32 + l31] = vval;
      if (l31 == 0) kreg[kk] = rid;
    }
    __syncthreads();

    // ---- S^T = bias + K Q^T : the bias is loaded straight into the accumulators (C-in) ------------
    f32x16 st[NKT];
    {
      // uniform base (SGPR) + one 32-bit lane offset; the per-register key offset is a constant soffset
      const int cbase = (int)(((long long)head * nk + (long long)c * KC) * 256 * 4);
      const int voff = (4 * hh * 256 + qi) * 4;
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kc = t * 32 + (r & 3) + 8 * (r >> 2);          // key row of lane half 0
          const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(brsrc, voff, cbase + kc * 1024, 0);
          st[t][r] = (c * KC + kc + 4 * hh < nk) ? __builtin_bit_cast(float, u) : 0.f;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
      for (int s = 0; s < 16; ++s)
        st[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[(2 * s + hh) * KTS + t * 32 + l31], qv[s], st[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- + mask, chunk max --------------------------------------------------------------------
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        float v = st[t][r];
        if (p.use_mask && kreg[kk] != qreg_id) v += -100.0f;
        if (c * KC + kk >= nk) v = -INFINITY;
        st[t][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float corr = expf(m_run - m_new);      // 0 on the first chunk (m_run = -inf)
    float ls = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = expf(st[t][r] - m_new);
        st[t][r] = e;
        ls += e;
      }
    }
    l_run = l_run * corr + ls;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= corr;
    __builtin_amdgcn_sched_barrier(0);
    // ---- O^T += V^T P^T : each S^T accumulator register is one k-step's B operand -----------------
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kk * 32 + l31], st[t][r], o, 0, 0, 0);
      }
    }
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (qvalid) {
    float* op = p.out + qtok * p.ldo + p.o_off + head * p.d;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (dd < p.d) op[dd] = o[r] * inv;
    }
  }
}

extern "C" int ff_window_attn(const float* qkv, int ldq, int q_off, int k_off, int v_off, float* out, int ldo,
                              int o_off, const float* biasT, int B, int H, int W, int Hp, int Wp, int wh, int ww,
                              int kh, int kw, int shift_h, int shift_w, int use_mask, int heads, int d,
                              float scale, void* stream) {
  FF_CHECK_ARG(qkv && out && biasT, "ff_window_attn: null pointer");
  FF_CHECK_ARG(wh * ww == 256, "ff_window_attn: query window must hold 256 tokens (got %dx%d)", wh, ww);
  FF_CHECK_ARG(d > 0 && d <= 32 && heads > 0, "ff_window_attn: head dim %d unsupported (<=32)", d);
  FF_CHECK_ARG(kh >= wh && kw >= ww && (kh - wh) % 2 == 0 && (kw - ww) % 2 == 0, "ff_window_attn: bad key window");
  FF_CHECK_ARG(Hp % wh == 0 && Wp % ww == 0 && Hp >= H && Wp >= W, "ff_window_attn: padded dims must tile by the window");
  FF_CHECK_ARG(shift_h >= 0 && shift_w >= 0 && shift_h < wh && shift_w < ww, "ff_window_attn: bad shift");
  FF_CHECK_ARG(!(shift_h || shift_w) || (kh == wh && kw == ww), "ff_window_attn: shift with overlapping keys unsupported");
  FF_CHECK_ARG(!use_mask || (shift_h > 0 && shift_w > 0), "ff_window_attn: mask needs a shift");
  const int nk = kh * kw;
  AttnParams p;
  p.qkv = qkv; p.out = out; p.biasT = biasT; p.ldq = ldq; p.ldo = ldo;
  p.q_off = q_off; p.k_off = k_off; p.v_off = v_off; p.o_off = o_off;
  p.B = B; p.H = H; p.W = W; p.Hp = Hp; p.Wp = Wp; p.wh = wh; p.ww = ww; p.kh = kh; p.kw = kw;
  p.sh = shift_h; p.sw = shift_w; p.use_mask = use_mask; p.heads = heads; p.d = d; p.scale = scale;
  p.nwx = Wp / ww; p.nwy = Hp / wh;
  const long long nblk = (long long)B * p.nwx * p.nwy * heads;
  FF_CHECK_ARG(nblk < (1LL << 31), "ff_window_attn: grid too large");
  hipStream_t st = (hipStream_t)stream;
  constexpr int NKT = 4;   // 128-key chunks: 64 accumulator registers, 2 waves/SIMD without in-loop spills
  const size_t lds = (size_t)(32 * (NKT * 32 + 1) + NKT * 32 * 32) * 4 + NKT * 32 * 4;
  hipLaunchKernelGGL(window_attn_kernel<NKT>, dim3((unsigned)nblk), dim3(512), lds, st, p);
  FF_LAUNCH_CHECK("ff_window_attn");
  return FF_OK;
}

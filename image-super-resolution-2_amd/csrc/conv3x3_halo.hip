// 3x3 / stride 1 / pad 1 convolution with the input tile resident in LDS ("halo" kernel), split-bf16 MFMA.
//   hat_arch.py:121-130 (CAB conv 180->60->180), :768 (RHAG conv), :921-953 (conv_after_body / upsample trunk),
//   dat_arch.py:396,772 (AIM / RG convs), nafnet_arch.py:187,193 (intro / ending), the 3x3 convs of the fusion stack.
// The generic implicit GEMM (conv_gemm_bf16.hip) gathers, converts and stages every input element once PER FILTER TAP
// (9x), which is what bounds it (~80 us for the 12.7 GFLOP CAB convs whose MFMA and HBM times are both ~15 us).
// Here a workgroup owns a TH x 16 pixel tile: the (TH+2) x 18 halo tile of 64 input channels is loaded, split into
// bf16 hi/lo and written to LDS ONCE, and the nine taps read it at shifted pixel rows -- the im2col is pure LDS
// addressing.  Weights arrive pre-split and pre-tiled (prep.pack_conv3x3_halo) as one LDS image per (n-block,
// 64-channel chunk, tap) and stream through a two-slot ring by LDS-DMA, overlapped with the 4 k-steps of the
// previous tap.  Accumulators stay in registers over all chunks and taps (K = 9*Cin).
//   LDS row (pixel or weight row) = [64 k hi | 64 k lo | 16 B pad] = 272 B = 68 dwords: conflict-free ds_read_b128.
#include "ff_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct HaloParams {
  const float* in; const unsigned char* w; const float* bias; const float* mul; const float* res; float* out;
  int B, H, W, Cin, ldi, Cout, ldo, ldr, nchunk, tiles_x, tiles_y, nblk, act, shuffle;
  float alpha;
  float* pool_part;   // optional [workgroups][Cout padded to nblk*BN]: per-workgroup channel sums of the stored values
  int vec_epi;        // 16-byte epilogue through LDS patches (set by the launcher when the layout allows)
  int io_bf16;        // nterms == 1 only: bit 0 = input rows are bf16 (ldi in elements), bit 1 = output rows are bf16 (ldo in elements)
#ifdef HX_TIMING
  unsigned long long* dbg;   // tools/hx_time.cpp: [block][wave][8] wall-clock stamps (debug build only)
#endif
};
#ifdef HX_TIMING
static unsigned long long* g_hx_dbg = nullptr;
#define HX_T(i) do { if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[((long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define HX_T(i) do { } while (0)
#endif

#define HX_ROW 272
#define HX_W 18
// Bytes per halo-tile ROW of 18 pixels, padded to 1280 dwords == 0 (mod 64 banks): a 32-pixel m-tile is two pixel rows, and
// ds_read_b128 services lanes {0-3, 12-15, 20-27} together -- with the natural pitch (18 x 68 = 1224 dwords == 8 mod 64) the second
// row's lanes landed on banks the first row's lanes 12-15 use (SQ_LDS_BANK_CONFLICT = 23 % of the LDS cycles, round 2).
#define HX_PROW 5120

// WK = input channels per weight tile (64: one tile per (chunk, tap); 32: two).  The 192-channel configuration uses
// 16x16-pixel workgroups (weights are re-streamed per workgroup: 256 pixels per fetch halve the L2 traffic that bounds
// the 128-pixel form) and 32-channel weight tiles so that the 88 KB input tile and the ring still fit in 160 KB.
// NTERMS = 3: split-bf16 products (fp32-grade).  NTERMS = 1: plain bf16 with a COMPACT LDS image -- pixel and weight rows hold the hi
// plane only (64 k x 2 B + 16 B pad = 144 B: 36 dwords per row puts the 16 rows of a ds_read_b128 lane group on 16 distinct 4-bank
// slots), halo-tile rows of 18 pixels are padded to 2816 B (== 0 mod 64 banks, as HX_PROW).  The 16x16-pixel tile then takes 50 KB
// instead of 92 KB, so the 64-channel configurations run TWO workgroups per CU: one's input staging / epilogue overlaps the other's taps
// (with one workgroup per CU those phases are serial: 35 us for the 12.7 GFLOP CAB convolutions against 7 us of MFMA time).
template <int N> __device__ __forceinline__ void halo_wait_vmcnt() {
  static_assert(N >= 0 && N <= 12, "counted vmcnt out of range");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

template <int NTERMS> struct HaloGeom {
  static constexpr int ROWB = NTERMS == 1 ? 144 : HX_ROW;          // bytes per pixel row in LDS
  static constexpr int PROW = NTERMS == 1 ? 2816 : HX_PROW;        // bytes per halo-tile row of 18 pixels
};
template <int WM, int WN, int MI, int NI, int WK, int ACT, int NSLOT, int NTERMS>
__global__ __launch_bounds__(WM * WN * 64, (NTERMS == 1 && MI * NI <= 2 && WM * WN == 8 && NSLOT != 9) ? 4 : (NTERMS == 1 && MI * NI <= 2 && WM * WN == 4 && WM == 4) ? 3 : (NTERMS == 1 && MI * NI == 6 && WM * WN == 4) ? 2 : 1)
void conv3x3_halo_kernel(HaloParams p) {
  constexpr int ROWB = HaloGeom<NTERMS>::ROWB, PROW = HaloGeom<NTERMS>::PROW;
  constexpr int NW = WM * WN, NT = NW * 64;            // 4 or 8 waves
  static_assert(NW == 4 || NW == 8, "four or eight waves");
  constexpr int PPP = NT / 16;                         // pixels staged per pass (16 lanes per pixel)
  constexpr int TH = WM * MI * 2, NPX = (TH + 2) * HX_W, NPASS = (NPX + PPP - 1) / PPP;
  constexpr int BN = WN * NI * 32, TN = NI * 32;
  constexpr int XBYTES = (TH + 2) * PROW;
  constexpr int WROW = (NTERMS == 1 ? WK * 2 : WK * 4) + 16, NH = 64 / WK, KSTEPS = WK / 16;
  constexpr bool BLK = NSLOT == 9;           // blocked schedule: a chunk's nine weight tiles resident, two barriers per chunk (below)
  constexpr bool TWO_PER_CU = NTERMS == 1 && MI * NI <= 2 && WM * WN == 8 && !BLK;   // 128-register budget: the partner workgroup hides the chunk load instead
  constexpr bool PAIR_4W = NTERMS == 1 && MI * NI == 6 && WM * WN == 4;      // 8 x 16 pixels x 192: two 4-wave workgroups per CU, 256 registers
  constexpr bool XPREF = MI * NI <= 6 && !TWO_PER_CU && !PAIR_4W;   // prefetch the next chunk's input rows into registers during tap 7
  constexpr int WPIECES = (BN * WROW + 1023) / 1024, WSLOT = WPIECES * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Xs = smem;
  unsigned char* Ws = smem + XBYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wr = wid / WN, wc = wid % WN;

  const int ntile = p.tiles_x * p.tiles_y * p.B;
  const int L = ff_xcd_remap(blockIdx.x, ntile * p.nblk);
  const int nb = L % p.nblk;
  int t = L / p.nblk;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = nb * BN;

  // ---- weight ring: tile index T = (chunk*9 + tap)*NH + half, image [nblk][nchunk*9*NH][WSLOT] -------------------
  const unsigned char* wimg = p.w + (long long)nb * p.nchunk * 9 * NH * WSLOT;
  auto dma = [&](int T, int slot) {
    const unsigned char* src = wimg + (long long)T * WSLOT + lane * 16;
#pragma unroll
    for (int i = 0; i < (WPIECES + NW - 1) / NW; ++i) {
      const int pc = wid + NW * i;
      if (pc < WPIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 1024),
                                         (__attribute__((address_space(3))) void*)(Ws + slot * WSLOT + pc * 1024), 16, 0, 0);
    }
  };
  const int ntiles = p.nchunk * 9 * NH;
  HX_T(0);
  // Ring of NSLOT weight slots, DEPTH = NSLOT - 1 tiles in flight: tile T + DEPTH is issued at the top of tile T (into the slot tile T - 1
  // just vacated) and waited for DEPTH tiles later with a COUNTED s_waitcnt vmcnt(NPMIN * (DEPTH - 1)) (NPMIN = the fewest pieces any wave
  // issues per tile; waves that issue one more over-wait by a piece: vmcnt retires in issue order) followed by a RAW s_barrier.
  // __syncthreads() would not do: with an LDS-DMA outstanding its fence waits vmcnt(0) and drains the ring (round 2 measured "no gain"
  // from three slots for exactly that reason).  With one tile in flight (NSLOT = 2) every tap exposes the L2 latency of its weight tile
  // (~0.7 us against 0.12 us of MFMA work per tap in plain bf16).
  constexpr int DEPTH = NSLOT - 1;
  constexpr int NPMIN = WPIECES / NW;
  if constexpr (BLK) {
#pragma unroll
    for (int tp = 0; tp < 5; ++tp) dma(tp, tp);
  } else {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (d < ntiles) dma(d, d);
  }

  // ---- input halo staging: 16 lanes (float4 each) per pixel, 16 pixels per pass ---------------------------------
  const int cq = (tid & 15) * 4, prow = tid >> 4;
  int goff[NPASS];
#pragma unroll
  for (int j = 0; j < NPASS; ++j) {
    const int hp = j * PPP + prow;
    const int hy = hp / HX_W, hx = hp - hy * HX_W;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool ok = hp < NPX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    goff[j] = ok ? ((b * p.H + iy) * p.W + ix) * p.ldi + cq : -1;
  }
  f32x4 xr[NPASS];
  auto load_x = [&](int chunk) {
    const int c0 = chunk * 64;
    const bool cok = c0 + cq < p.Cin;
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      const bool ok = cok && goff[j] >= 0;
      f32x4 u;
      if (NTERMS == 1 && (p.io_bf16 & 1)) {                   // bf16 rows: four values are 8 bytes; the conversion back in store_x is exact
        const bf16x4 h4 = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.in) + (ok ? goff[j] + c0 : 0));
        u = (f32x4){(float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]};
      } else {
        u = *reinterpret_cast<const f32x4*>(p.in + (ok ? goff[j] + c0 : 0));
      }
      xr[j] = ok ? u : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_x = [&]() {
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      const int hp = j * PPP + prow;
      bf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float f = xr[j][e];
        const __bf16 h = (__bf16)f;
        hi[e] = h;
        lo[e] = (__bf16)(f - (float)h);
      }
      if (hp < NPX) {
        unsigned char* dst = Xs + (hp / HX_W) * PROW + (hp % HX_W) * ROWB + (tid & 15) * 8;
        *reinterpret_cast<bf16x4*>(dst) = hi;
        if (NTERMS == 3) *reinterpret_cast<bf16x4*>(dst + 128) = lo;
      }
    }
  };
  load_x(0);
  store_x();
  HX_T(1);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane LDS byte offsets (tap (0,0)): pixel row of m-tile i, weight row of n-tile j
  int aoff[MI], boff[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) aoff[i] = ((wr * MI + i) * 2 + (l31 >> 4)) * PROW + (l31 & 15) * ROWB + 16 * hh;
#pragma unroll
  for (int j = 0; j < NI; ++j) boff[j] = (wc * TN + j * 32 + l31) * WROW + 16 * hh;

  // one tap of the current chunk from weight slot `slot` (all KSTEPS k-steps)
  auto tap_mfma = [&](int tap, int slot) {
    const int dy = tap / 3, dx = tap - 3 * dy;
    const unsigned char* xa = Xs + dy * PROW + dx * ROWB;
    const unsigned char* wb = Ws + slot * WSLOT;
    // every operand of the tap is requested before the first MFMA: 12 ds_read_b128 in flight instead of 3 per k-step
    bf16x8 ah[KSTEPS][MI], bh[KSTEPS][NI];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
      for (int i = 0; i < MI; ++i) ah[s][i] = *reinterpret_cast<const bf16x8*>(xa + aoff[i] + 32 * s);
#pragma unroll
      for (int j = 0; j < NI; ++j) bh[s][j] = *reinterpret_cast<const bf16x8*>(wb + boff[j] + 32 * s);
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][i], bh[s][j], acc[i][j], 0, 0, 0);
  };
  if constexpr (BLK) {
    // Blocked schedule (plain bf16, 64-channel weight tiles, one workgroup per CU): the nine tap tiles of a chunk have fixed slots; taps
    // 0..4 are computed while taps 5..8 land, taps 5..8 while the next chunk's 0..4 (and its input rows, in registers) land -- three
    // barriers per chunk instead of nine, 40 / 32 MFMAs per wave between them instead of 8 (tools/hx_time.cpp: 0.7 us per tap of the
    // ring form against 0.1 us of MFMA issue).
    static_assert(NTERMS == 1 && NH == 1, "blocked schedule: plain bf16, one weight tile per tap");
    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // taps 0..4 and the input tile of this chunk are in LDS
#pragma unroll
      for (int tp = 5; tp < 9; ++tp) dma(chunk * 9 + tp, tp);
#pragma unroll
      for (int tp = 0; tp < 5; ++tp) tap_mfma(tp, tp);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // taps 5..8 landed; slots 0..4 are free
      if (chunk + 1 < p.nchunk) {
#pragma unroll
        for (int tp = 0; tp < 5; ++tp) dma((chunk + 1) * 9 + tp, tp);
        load_x(chunk + 1);
      }
#pragma unroll
      for (int tp = 5; tp < 9; ++tp) tap_mfma(tp, tp);
      if (chunk + 1 < p.nchunk) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // every wave is done reading this chunk's input tile
        store_x();
      }
    }
  } else {
  int T = 0;
    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - 3 * dy;
        const unsigned char* xa = Xs + dy * PROW + dx * ROWB;
  #pragma unroll
        for (int half = 0; half < NH; ++half, ++T) {
          // own DMA pieces of tile T have landed (counted wait: the younger tiles stay in flight); LDS writes of this wave (the staged
          // input tile) are complete; the barrier then makes every wave's pieces and the input tile visible, and guarantees the slot of
          // tile T - 1 is no longer being read
          if (T + DEPTH - 1 < ntiles) halo_wait_vmcnt<NPMIN * (DEPTH - 1)>();
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (T + DEPTH < ntiles) dma(T + DEPTH, (T + DEPTH) % NSLOT);
          if (XPREF && tap == 7 && half == NH - 1 && chunk + 1 < p.nchunk) load_x(chunk + 1);
          const unsigned char* wb = Ws + (T % NSLOT) * WSLOT;
  #pragma unroll
          for (int s = 0; s < KSTEPS; ++s) {
            bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
  #pragma unroll
            for (int i = 0; i < MI; ++i) {
              ah[i] = *reinterpret_cast<const bf16x8*>(xa + aoff[i] + 32 * (half * KSTEPS + s));
              if (NTERMS == 3) al[i] = *reinterpret_cast<const bf16x8*>(xa + aoff[i] + 32 * (half * KSTEPS + s) + 128);
            }
  #pragma unroll
            for (int j = 0; j < NI; ++j) {
              bh[j] = *reinterpret_cast<const bf16x8*>(wb + boff[j] + 32 * s);
              if (NTERMS == 3) bl[j] = *reinterpret_cast<const bf16x8*>(wb + boff[j] + 32 * s + WK * 2);
            }
  #pragma unroll
            for (int i = 0; i < MI; ++i)
  #pragma unroll
              for (int j = 0; j < NI; ++j) {
                if (NTERMS == 3) {
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
              }
          }
        }
      }
      if (chunk + 1 < p.nchunk) {
        __syncthreads();                 // every wave is done reading this chunk's input tile
        if (!XPREF) load_x(chunk + 1);   // big-accumulator configuration: no registers to hold the prefetch across the taps
        store_x();                       // (made visible by the barrier at the top of the next tap)
      }
    }
  }
  HX_T(2);
  // ---- vector epilogue (no pixel shuffle, Cout % 4 == 0, 16-byte aligned rows): every 32 x 32 accumulator tile goes through the wave's
  //      LDS patch so that a lane owns four consecutive channels of a pixel -- four 16-byte stores of whole 128-byte pixel segments per
  //      tile instead of sixteen 4-byte stores (tools/hx_time.cpp: the scalar epilogue was 10.7 us of CAB 60->180's 32 us and did not
  //      shrink when the output went to bf16: it is bound by store instructions, not by bytes) ------------------------------------------
  if (p.vec_epi) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // every wave has left the tap loop: the staging area is free
    float* ps = reinterpret_cast<float*>(smem);          // [WM][BN] pool partials (first 8 KB), then the waves' 32 x 36 patches
    float* patch = reinterpret_cast<float*>(smem + 8192) + wid * (32 * 36);
    const int tq = lane >> 3, q4 = 4 * (lane & 7);
    const bool has_res = p.res != nullptr;
    // residual quads are requested one n-tile ahead of their use (fetched right before the add, each tile exposed an HBM latency: the
    // finding of sgfn_tail's epilogue; all tiles at once spills the big-accumulator forms)
    constexpr bool RES_AHEAD = !(TWO_PER_CU || (NTERMS == 1 && MI * NI <= 2 && WM * WN == 4));   // the 128- / 168-register forms fetch at use
    f32x4 rq[2][MI][4];
    auto load_res = [&](int j, f32x4 (&dst)[MI][4]) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int mrow = tq + 8 * k, c4 = n0 + wc * TN + j * 32 + q4;
          const int oy = y0 + (wr * MI + i) * 2 + (mrow >> 4), ox = x0 + (mrow & 15);
          const bool ok = c4 < p.Cout && oy < p.H && ox < p.W;
          dst[i][k] = ok ? *reinterpret_cast<const f32x4*>(p.res + ((long long)(b * p.H + oy) * p.W + ox) * p.ldr + c4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    if (RES_AHEAD && has_res) load_res(0, rq[0]);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wc * TN + j * 32 + l31;
      const bool nok = n < p.Cout;
      const float bv = (nok && p.bias) ? p.bias[n] : 0.f;
      const float mv = ((nok && p.mul) ? p.mul[n] : 1.f) * p.alpha;
      const int c4 = n0 + wc * TN + j * 32 + q4;         // the four channels this lane stores
      const bool cok = c4 < p.Cout;
      if (RES_AHEAD && has_res && j + 1 < NI) load_res(j + 1, rq[(j + 1) & 1]);
      f32x4 mv4 = {p.alpha, p.alpha, p.alpha, p.alpha};
      if (p.mul && cok) mv4 = *reinterpret_cast<const f32x4*>(p.mul + c4) * p.alpha;
      float psum = 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ry0 = y0 + (wr * MI + i) * 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {                   // register r = tile row (r & 3) + 8 (r >> 2) + 4 hh of channel l31
          const int mrow = (r & 3) + 8 * (r >> 2) + 4 * hh;
          const float a = ff_act_c<ACT, true>(acc[i][j][r] + bv);
          patch[mrow * 36 + l31] = a;                      // scale and residual are applied after the transposition, as one fma (the scalar path's rounding)
          if (nok && ry0 + (mrow >> 4) < p.H && x0 + (mrow & 15) < p.W) psum += a * mv;  // (pool partials: no residual on this path)
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int mrow = tq + 8 * k;
          const int oy = ry0 + (mrow >> 4), ox = x0 + (mrow & 15);
          const bool ok = cok && oy < p.H && ox < p.W;
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(patch + mrow * 36 + q4);
          const long long pix = ok ? ((long long)(b * p.H + oy) * p.W + ox) : 0;
          f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
          if (RES_AHEAD && has_res) r4 = rq[j & 1][i][k];
          else if (has_res && ok) r4 = *reinterpret_cast<const f32x4*>(p.res + pix * p.ldr + c4);
          f32x4 v4;
#pragma unroll
          for (int e = 0; e < 4; ++e) v4[e] = __builtin_fmaf(a4[e], mv4[e], r4[e]);
          if (ok) {
            if (NTERMS == 1 && (p.io_bf16 & 2)) {
              bf16x4 h4;
#pragma unroll
              for (int e = 0; e < 4; ++e) h4[e] = (__bf16)v4[e];
              *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out) + pix * p.ldo + c4) = h4;
            } else {
              *reinterpret_cast<f32x4*>(p.out + pix * p.ldo + c4) = v4;
            }
          }
        }
      }
      if (p.pool_part) {
        psum += __shfl_xor(psum, 32);
        if (hh == 0) ps[wr * BN + wc * TN + j * 32 + l31] = psum;
      }
    }
    if (p.pool_part) {
      __syncthreads();
      for (int c = threadIdx.x; c < BN; c += NT) {
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) sacc += ps[w * BN + c];
        p.pool_part[(long long)blockIdx.x * BN + c] = sacc;
      }
    }
  } else
  // ---- epilogue: lane = output channel column, 16 pixels per m-tile ---------------------------------------------
  {
    const bool has_res = p.res != nullptr;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wc * TN + j * 32 + l31;
      const bool nok = n < p.Cout;         // (no early `continue`: every wave reaches the pooling barriers below)
      const int nc = nok ? n : 0;
      const float bv = (nok && p.bias) ? p.bias[n] : 0.f;
      const float mv = ((nok && p.mul) ? p.mul[n] : 1.f) * p.alpha;
      float psum = 0.f;                    // this lane's channel, summed over the pixels of the wave's m-tiles
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        long long oidx[16];
        float rv[16];
        bool okp[16];
        const int ry0 = y0 + (wr * MI + i) * 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mrow = (r & 3) + 8 * (r >> 2) + 4 * hh;
          const int oy = ry0 + (mrow >> 4), ox = x0 + (mrow & 15);
          okp[r] = nok && oy < p.H && ox < p.W;
          if (p.shuffle == 2) {              // fused PixelShuffle(2): channel n = 4 co + 2 sy + sx -> pixel (2 oy + sy, 2 ox + sx)
            const int co = nc >> 2, sy = (nc >> 1) & 1, sx = nc & 1;
            const long long pix = okp[r] ? ((long long)(b * 2 * p.H + 2 * oy + sy) * (2 * p.W) + 2 * ox + sx) : 0;
            oidx[r] = pix * p.ldo + co;
            rv[r] = has_res ? p.res[pix * p.ldr + co] : 0.f;
          } else {
            const long long pix = okp[r] ? ((long long)(b * p.H + oy) * p.W + ox) : 0;
            oidx[r] = pix * p.ldo + nc;
            rv[r] = has_res ? p.res[pix * p.ldr + nc] : 0.f;
          }
        }
        if (ry0 + 1 < p.H && x0 + 15 < p.W) {
          if (nok) {                         // one exec mask for the whole tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = __builtin_fmaf(ff_act_c<ACT, true>(acc[i][j][r] + bv), mv, rv[r]);
              if (NTERMS == 1 && (p.io_bf16 & 2)) reinterpret_cast<__bf16*>(p.out)[oidx[r]] = (__bf16)v;
              else p.out[oidx[r]] = v;
              psum += v;
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (okp[r]) {
              const float v = __builtin_fmaf(ff_act_c<ACT, true>(acc[i][j][r] + bv), mv, rv[r]);
              if (NTERMS == 1 && (p.io_bf16 & 2)) reinterpret_cast<__bf16*>(p.out)[oidx[r]] = (__bf16)v;
              else p.out[oidx[r]] = v;
              psum += v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // one tile at a time: do not interleave the address / residual work of all MI*NI tiles
      }
      // global-average-pool partials of the output (hat_arch.py:50 ChannelAttention pools exactly this tensor): the lane's
      // channel over its wave's pixels -> both lane halves -> the WM waves of the workgroup (LDS) -> one value per channel
      if (p.pool_part) {
        psum += __shfl_xor(psum, 32);
        float* ps = reinterpret_cast<float*>(smem);              // [WM][BN]: staging buffers are idle in the epilogue
        if (j == 0) __syncthreads();                              // (first use: every wave has left the tap loop)
        if (hh == 0) ps[wr * BN + wc * TN + j * 32 + l31] = psum;
      }
    }
    if (p.pool_part) {
      __syncthreads();
      const float* ps = reinterpret_cast<const float*>(smem);
      for (int c = threadIdx.x; c < BN; c += NT) {
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) sacc += ps[w * BN + c];
        p.pool_part[(long long)blockIdx.x * BN + c] = sacc;
      }
    }
  }
  HX_T(3);
#ifdef HX_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  HX_T(4);
#endif
}

template <int WM, int WN, int MI, int NI, int WK, int NSLOT, int NTERMS>
static int launch_halo(HaloParams& p, hipStream_t st) {
  constexpr int TH = WM * MI * 2, BN = WN * NI * 32;
  constexpr int WSLOT = ((BN * ((NTERMS == 1 ? WK * 2 : WK * 4) + 16) + 1023) / 1024) * 1024;
  constexpr size_t lds = (size_t)(TH + 2) * HaloGeom<NTERMS>::PROW + NSLOT * WSLOT;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static_assert(lds >= 8192 + (size_t)WM * WN * 32 * 36 * 4, "the vector epilogue's pool row + patches must fit in the staging area");
  p.tiles_x = (p.W + 15) / 16;
  p.tiles_y = (p.H + TH - 1) / TH;
  p.nblk = (p.Cout + BN - 1) / BN;
  const long long nblocks = (long long)p.tiles_x * p.tiles_y * p.B * p.nblk;
  if (nblocks >= (1LL << 31)) { ff_set_error("ff_conv3x3_halo: grid too large"); return FF_ERR_ARG; }
  bool attr_failed = false;
  auto go = [&](auto A) {                      // the activation is a template parameter: one epilogue per kernel
    constexpr int ACT = decltype(A)::value;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<WM, WN, MI, NI, WK, ACT, NSLOT, NTERMS>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { ff_set_error("ff_conv3x3_halo: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); attr_failed = true; return; }
      attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_halo_kernel<WM, WN, MI, NI, WK, ACT, NSLOT, NTERMS>), dim3((unsigned)nblocks), dim3(WM * WN * 64), lds, st, p);
  };
  FF_DISPATCH_ACT(p.act, go);
  if (attr_failed) return FF_ERR_LAUNCH;
  FF_LAUNCH_CHECK("ff_conv3x3_halo");
  return FF_OK;
}

// bytes of the weight image prep.pack_conv3x3_halo must produce for (Cout, Cin, bn)
extern "C" long long ff_conv3x3_halo_weight_bytes(int Cout, int Cin, int bn, int nterms) {
  if (Cout <= 0 || Cin <= 0 || (bn != 32 && bn != 64 && bn != 128 && bn != 192) || (nterms != 1 && nterms != 3)) return -1;
  const int wk = bn == 192 ? 32 : 64;                        // channels per weight tile (see launch_halo)
  const long long slot = ((long long)bn * ((nterms == 1 ? wk * 2 : wk * 4) + 16) + 1023) / 1024 * 1024;
  return (long long)((Cout + bn - 1) / bn) * ((Cin + 63) / 64) * 9 * (64 / wk) * slot;
}

// number of workgroups ( = rows of pool_partials, each nblk... see below) for (B, H, W, Cout, bn): rows x row length
// pixel rows per workgroup (the launch table below).  Plain bf16, 32 / 64 output channels, and fewer than two 16x16 tiles per CU
// (the 256x256 token grids of HAT / DAT: 256 tiles): 8-row tiles of four waves, 46 KB of LDS, up to three workgroups per CU -- with
// one 16-row workgroup per CU the launch is as long as ONE workgroup's serial load / taps / store sequence.
static int halo_rows_per_wg(int B, int H, int W, int Cout, int bn, int nterms) {
  if (bn == 128) return 8;
  static const int blocked = []() { const char* e = getenv("FF_HALO_BLOCKED"); return e ? atoi(e) : 1; }();   // tuning switch
  if (nterms == 1 && bn == 64 && blocked) {
    const long long tiles16 = (long long)B * ((H + 15) / 16) * ((W + 15) / 16) * ((Cout + bn - 1) / bn);
    if (tiles16 < 512 || blocked == 2) return 16;        // blocked 16-row form (launch table below)
  }
  static const int wide8 = []() { const char* e = getenv("FF_HALO_192_8ROW"); return e ? atoi(e) : 0; }();   // tuning switch: 8-row form for 192 outputs (measured 36.9 vs 35.8 us on CAB 60->180, 67.1 vs 65.7 on 180->180: off)
  if (nterms == 1 && (bn <= 64 || (bn == 192 && wide8))) {
    const long long tiles16 = (long long)B * ((H + 15) / 16) * ((W + 15) / 16) * ((Cout + bn - 1) / bn);
    if (tiles16 < 512) return 8;
  }
  return 16;
}

extern "C" long long ff_conv3x3_halo_pool_rows(int B, int H, int W, int Cout, int bn, int nterms) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (bn != 32 && bn != 64 && bn != 128 && bn != 192) || (nterms != 1 && nterms != 3)) return -1;
  const int th = halo_rows_per_wg(B, H, W, Cout, bn, nterms);
  return (long long)B * ((H + th - 1) / th) * ((W + 15) / 16) * ((Cout + bn - 1) / bn);
}

extern "C" int ff_conv3x3_halo(const float* in, int ldi, const void* w_img, int bn, const float* bias, const float* mul,
                               const float* res, int ldr, float* out, int ldo, int B, int H, int W, int Cin, int Cout,
                               int act, float alpha, int shuffle, float* pool_partials, int nterms, int io_bf16, void* stream) {
  FF_CHECK_ARG(nterms == 1 || nterms == 3, "ff_conv3x3_halo: nterms must be 1 or 3");
  FF_CHECK_ARG(io_bf16 == 0 || (nterms == 1 && io_bf16 > 0 && io_bf16 < 4 && shuffle == 0 && !res), "ff_conv3x3_halo: bf16 input / output rows exist for nterms == 1, no shuffle, no residual");
  FF_CHECK_ARG(in && w_img && out, "ff_conv3x3_halo: null pointer");
  FF_CHECK_ARG(!pool_partials || (shuffle == 0 && Cout <= bn), "ff_conv3x3_halo: pool partials need Cout <= bn and no pixel shuffle");
  FF_CHECK_ARG(shuffle == 0 || (shuffle == 2 && Cout % 4 == 0), "ff_conv3x3_halo: shuffle must be 0 or 2 (Cout %% 4 == 0)");
  FF_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "ff_conv3x3_halo: bad dims");
  FF_CHECK_ARG(Cin % 4 == 0 && ldi % 4 == 0 && ldi >= Cin && (((uintptr_t)in) & 15) == 0, "ff_conv3x3_halo: input rows must be 16-byte aligned (8-byte for bf16 rows), Cin %% 4 == 0");
  FF_CHECK_ARG((((uintptr_t)w_img) & 15) == 0, "ff_conv3x3_halo: weight image must be 16-byte aligned");
  FF_CHECK_ARG(ldo >= (shuffle ? Cout / 4 : Cout) && (!res || ldr >= (shuffle ? Cout / 4 : Cout)), "ff_conv3x3_halo: ldo / ldr too small");
  FF_CHECK_ARG((long long)B * H * W * ldi < (1LL << 31), "ff_conv3x3_halo: input too large for 32-bit offsets");
  FF_CHECK_ARG(in != out, "ff_conv3x3_halo: in-place convolution is not supported");
  HaloParams p;
  p.in = in; p.w = (const unsigned char*)w_img; p.bias = bias; p.mul = mul; p.res = res; p.out = out;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.ldi = ldi; p.Cout = Cout; p.ldo = ldo; p.ldr = ldr;
  p.nchunk = (Cin + 63) / 64; p.act = act; p.alpha = alpha; p.shuffle = shuffle; p.pool_part = pool_partials; p.io_bf16 = io_bf16;
  static const int vec_epi_on = []() { const char* e = getenv("FF_HALO_VEC_EPI"); return e ? atoi(e) : 1; }();   // tuning switch
  p.vec_epi = vec_epi_on && shuffle == 0 && Cout % 4 == 0 && ldo % 4 == 0 && (((uintptr_t)out) & 15) == 0 && !(res && pool_partials) &&
              (!res || (ldr % 4 == 0 && (((uintptr_t)res) & 15) == 0)) && (!mul || (((uintptr_t)mul) & 15) == 0);
#ifdef HX_TIMING
  p.dbg = g_hx_dbg;
#endif
  hipStream_t st = (hipStream_t)stream;
  static const int ring = []() { const char* e = getenv("FF_HALO_RING"); return e ? atoi(e) : 0; }();   // tuning switch: weight-ring slots
  static const int blocked = []() { const char* e = getenv("FF_HALO_BLOCKED"); return e ? atoi(e) : 1; }();
  if (nterms == 1 && bn == 64 && blocked) {
    const long long tiles16 = (long long)B * ((H + 15) / 16) * ((W + 15) / 16) * ((Cout + bn - 1) / bn);
    if (tiles16 < 512 || blocked == 2) return launch_halo<8, 1, 1, 2, 64, 9, 1>(p, st);    // nine resident tap tiles, three barriers per chunk
  }
  if (nterms == 1 && bn <= 64 && halo_rows_per_wg(B, H, W, Cout, bn, nterms) == 8) {   // 8x16 pixels, 4 waves
    if (bn == 32) return launch_halo<4, 1, 1, 1, 64, 2, 1>(p, st);
    if (ring == 3) return launch_halo<4, 1, 1, 2, 64, 3, 1>(p, st);
    if (ring == 4) return launch_halo<4, 1, 1, 2, 64, 4, 1>(p, st);
    if (ring == 5) return launch_halo<4, 1, 1, 2, 64, 5, 1>(p, st);
    return launch_halo<4, 1, 1, 2, 64, 2, 1>(p, st);
  }
  if (nterms == 1 && bn == 192 && halo_rows_per_wg(B, H, W, Cout, bn, nterms) == 8)
    return launch_halo<2, 2, 2, 3, 32, 2, 1>(p, st);      // 8x16 pixels x 192, 4 waves, 58 KB: two workgroups per CU on the 256 x 256 grids
  if (nterms == 1 && bn == 64 && ring == 3) return launch_halo<8, 1, 1, 2, 64, 3, 1>(p, st);
  if (nterms == 1 && bn == 192 && ring >= 3) return launch_halo<4, 2, 2, 3, 32, 3, 1>(p, st);
  switch (bn) {
    // 16x16-pixel workgroups of 8 waves (two per SIMD: one wave's LDS reads and waits hide behind the
    // other's MFMAs -- measured 140 -> 121 us for 180->180 against the 4-wave form)
    case 32: return nterms == 3 ? launch_halo<8, 1, 1, 1, 64, 2, 3>(p, st) : launch_halo<8, 1, 1, 1, 64, 2, 1>(p, st);    // wave: 32 pixels x 32 channels
    case 64: return nterms == 3 ? launch_halo<8, 1, 1, 2, 64, 2, 3>(p, st) : launch_halo<8, 1, 1, 2, 64, 2, 1>(p, st);    // wave: 32 pixels x 64
    case 128: return nterms == 3 ? launch_halo<2, 2, 2, 2, 64, 2, 3>(p, st) : launch_halo<2, 2, 2, 2, 64, 2, 1>(p, st);   // 8x16 pixels, 4 waves (only the two up-sampling convolutions use it)
    case 192: return nterms == 3 ? launch_halo<4, 2, 2, 3, 32, 2, 3>(p, st) : launch_halo<4, 2, 2, 3, 32, 2, 1>(p, st);   // 16x16 pixels x 192, 8 waves (2 per SIMD), 32-channel weight tiles; 88 KB tile + 2 x 27 KB ring
    default: ff_set_error("ff_conv3x3_halo: bn must be 32, 64, 128 or 192"); return FF_ERR_ARG;
  }
}

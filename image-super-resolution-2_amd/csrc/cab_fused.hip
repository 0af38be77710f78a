// HAT's convolution branch in ONE launch, plain bf16 (hat_arch.py:61-74 CAB: conv3x3 C -> C/3, GELU, conv3x3 C/3 -> C; the
// ChannelAttention that follows needs the global average pool of the result: emitted as per-workgroup partial sums, hat_arch.py:50).
//
// Why: in plain bf16 the two LDS-resident 3x3 launches (csrc/conv3x3_halo.hip) are ~40 us each in situ for 63 MB of traffic -- per-launch
// ramp, one serial load -> taps -> store sequence per workgroup, and the C/3-channel tensor written and read in between.  Here a
// workgroup owns a 16x16 tile of the OUTPUT: it evaluates the first convolution on the 18x18 halo of that tile (1.27x its work,
// recomputed, never stored), keeps GELU(conv1) as a bf16 image in LDS -- exactly the operand the second convolution would have formed from
// the fp32 tensor -- and runs the second convolution from it.  Results are bit-identical to the two-launch bf16 path (same K order, same
// fast GELU, same rounding points); only 47 MB in + 47 MB out (+ the 1.56x input halo) cross HBM.
//
// LDS (bf16, hi plane only; rows of 64 k x 2 B + 16 B pad = 144 B: conflict-free ds_read_b128 lane groups, see conv3x3_halo.hip):
//   X  image  20 x 20 pixels of one 64-channel chunk of the input          [20][XP = 2944]     58 880 B   (phase 1)
//   C1 image  18 x 18 pixels x 64 mid channels (60 used)                    [18][CP = 2816]     50 688 B
//   W  ring   phase 1: conv1 tiles 64 rows x 144 B per (chunk, tap); phase 2: conv2 tiles 192 rows x 80 B per (tap, half) in X's space
// Weight images are the nterms = 1 images of ff_conv3x3_halo (prep.pack_conv3x3_halo(w1, Cin, 64, 1) and (w2, Cmid, 192, 1)).
#include "ff_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct CabParams {
  const float* in; const unsigned char* w1; const float* b1; const unsigned char* w2; const float* b2; float* out; float* pool_part;
  int H, W, Cin, ldi, Cmid, Cout, ldo, nchunk, tiles_x, tiles_y;
};

#define CB_ROW 144
#define CB_XP 2944          // 20 pixels x 144 B = 2880, padded to a multiple of 128 B that keeps consecutive rows 8 banks apart
#define CB_CP 2816          // 18 pixels x 144 B = 2592, padded as the halo kernel's compact rows (== 0 mod 64 banks)
#define CB_XB (20 * CB_XP)
#define CB_CB (18 * CB_CP)
#define CB_W1SLOT 9216      // 64 rows x 144 B
#define CB_W2SLOT 15360     // 192 rows x 80 B
#define CB_NM1 11           // m-tiles of 32 positions covering the 18 x 18 = 324 halo positions (352)

template <int N> __device__ __forceinline__ void cab_wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
}

__global__ __launch_bounds__(512) void cab_fused_kernel(CabParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Xs = smem;                               // phase 1 input chunk; phase 2: conv2 weight ring
  unsigned char* Cs = smem + CB_XB + 2 * CB_W1SLOT;       // GELU(conv1) image
  unsigned char* W1s = smem + CB_XB;                      // conv1 weight ring (2 slots)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int L = ff_xcd_remap(blockIdx.x, p.tiles_x * p.tiles_y);
  const int tx = L % p.tiles_x, ty = L / p.tiles_x;
  const int y0 = ty * 16, x0 = tx * 16;

  // ------------------------------------------------------------------------------------------------ phase 1: conv1 on the halo
  auto dma1 = [&](int T, int slot) {                       // 9 pieces of 1 KiB, waves 0..7 then wave 0 again
    const unsigned char* src = p.w1 + (long long)T * CB_W1SLOT + lane * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wid + 8 * i;
      if (pc < 9)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 1024),
                                         (__attribute__((address_space(3))) void*)(W1s + slot * CB_W1SLOT + pc * 1024), 16, 0, 0);
    }
  };
  const int ntiles1 = p.nchunk * 9;
  dma1(0, 0);
  // input staging: 16 lanes (float4 each) per pixel, 32 pixels per pass, 400 pixels -> 13 passes
  const int cq = (tid & 15) * 4, prow = tid >> 4;
  int goff[13];
#pragma unroll
  for (int j = 0; j < 13; ++j) {
    const int hp = j * 32 + prow;
    const int hy = hp / 20, hx = hp - hy * 20;
    const int iy = y0 - 2 + hy, ix = x0 - 2 + hx;
    const bool ok = hp < 400 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    goff[j] = ok ? (iy * p.W + ix) * p.ldi + cq : -1;
  }
  // the next chunk's rows are requested at the top of the current chunk's taps and converted / written when the taps are done: the
  // HBM phase of chunks 1.. hides under the matrix work (with every workgroup of the launch in the same phase at the same time, a
  // load phase that nothing overlaps leaves the matrix pipes idle and a tap phase leaves HBM idle)
  f32x4 xr[13];
  auto load_x = [&](int chunk) {
    const int c0 = chunk * 64;
    const bool cok = c0 + cq < p.Cin;
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      const bool ok = cok && goff[j] >= 0;
      const f32x4 u = *reinterpret_cast<const f32x4*>(p.in + (ok ? goff[j] + c0 : 0));
      xr[j] = ok ? u : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_x = [&]() {
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      const int hp = j * 32 + prow;
      bf16x4 hi;
#pragma unroll
      for (int e = 0; e < 4; ++e) hi[e] = (__bf16)xr[j][e];
      if (hp < 400) *reinterpret_cast<bf16x4*>(Xs + (hp / 20) * CB_XP + (hp % 20) * CB_ROW + (tid & 15) * 8) = hi;
    }
  };
  load_x(0);
  store_x();

  // m-tiles of this wave: wid and wid + 8 (the second only for waves 0..2); position p = 32 mt + l31 of the 18 x 18 grid
  int aoff1[2];
  bool mt_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int mt = wid + 8 * i;
    mt_ok[i] = mt < CB_NM1;
    int pos = mt * 32 + l31;
    if (pos > 323) pos = 323;                              // padded rows read a valid position; their results are dropped
    aoff1[i] = (pos / 18) * CB_XP + (pos % 18) * CB_ROW + 16 * hh;
  }
  const int boff1 = l31 * CB_ROW + 16 * hh;               // n-tile j: + j * 32 rows
  f32x16 acc1[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;

  int T = 0;
  for (int chunk = 0; chunk < p.nchunk; ++chunk) {
    for (int tap = 0; tap < 9; ++tap, ++T) {
      const int dy = tap / 3, dx = tap - 3 * dy;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile T landed (and this wave's staged rows are written: lgkmcnt below)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (T + 1 < ntiles1) dma1(T + 1, (T + 1) & 1);
      if (tap == 1 && chunk + 1 < p.nchunk) load_x(chunk + 1);
      const unsigned char* xa = Xs + dy * CB_XP + dx * CB_ROW;
      const unsigned char* wb = W1s + (T & 1) * CB_W1SLOT;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(xa + aoff1[i] + 32 * s);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(wb + boff1 + j * 32 * CB_ROW + 32 * s);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j], acc1[0][j], 0, 0, 0);
        if (mt_ok[1]) {                                      // wave-uniform
#pragma unroll
          for (int j = 0; j < 2; ++j) acc1[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j], acc1[1][j], 0, 0, 0);
        }
      }
    }
    if (chunk + 1 < p.nchunk) {
      __syncthreads();                                      // every wave is done reading this chunk's input tile
      store_x();
    }
  }
  // conv2's first weight tiles can start now only after every wave has left the X image: they share its space
  __syncthreads();
  auto dma2 = [&](int T2, int slot) {                      // 15 pieces of 1 KiB
    const unsigned char* src = p.w2 + (long long)T2 * CB_W2SLOT + lane * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wid + 8 * i;
      if (pc < 15)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 1024),
                                         (__attribute__((address_space(3))) void*)(Xs + slot * CB_W2SLOT + pc * 1024), 16, 0, 0);
    }
  };
  dma2(0, 0);
  // GELU(conv1 + bias) -> bf16 -> C1 image; positions outside the image are ZERO (they are conv2's zero padding)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!mt_ok[i]) continue;
    const int mt = wid + 8 * i;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = j * 32 + l31;
      const float bv = n < p.Cmid ? p.b1[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pos = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (pos < 324) {
          const int py = pos / 18, px = pos - py * 18;
          const int iy = y0 - 1 + py, ix = x0 - 1 + px;
          const bool inside = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && n < p.Cmid;
          const float v = inside ? ff_act_c<ACT_GELU, true>(acc1[i][j][r] + bv) : 0.f;
          *reinterpret_cast<__bf16*>(Cs + py * CB_CP + px * CB_ROW + n * 2) = (__bf16)v;
        }
      }
    }
  }

  // ------------------------------------------------------------------------------------------------ phase 2: conv2 from the C1 image
  f32x16 acc2[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
  const int aoff2 = (wid * 2 + (l31 >> 4)) * CB_CP + (l31 & 15) * CB_ROW + 16 * hh;     // wave = two output rows
  const int boff2 = l31 * 80 + 16 * hh;
  for (int T2 = 0; T2 < 18; ++T2) {                         // (tap, half): 32 mid channels per tile
    const int tap = T2 >> 1, half = T2 & 1;
    const int dy = tap / 3, dx = tap - 3 * dy;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (T2 + 1 < 18) dma2(T2 + 1, (T2 + 1) & 1);
    const unsigned char* ca = Cs + dy * CB_CP + dx * CB_ROW + 64 * half;
    const unsigned char* wb = Xs + (T2 & 1) * CB_W2SLOT;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(ca + aoff2 + 32 * s);
      bf16x8 b[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) b[j] = *reinterpret_cast<const bf16x8*>(wb + boff2 + j * 32 * 80 + 32 * s);
#pragma unroll
      for (int j = 0; j < 6; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc2[j], 0, 0, 0);
    }
  }

  // ------------------------------------------------------------------------------------------------ epilogue: bias, store, pool partials
  __syncthreads();                                          // the staging space becomes the pooling scratch [8 waves][192]
  float* ps = reinterpret_cast<float*>(smem);
  const int ry0 = y0 + wid * 2;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int n = j * 32 + l31;
    const bool nok = n < p.Cout;
    const float bv = nok ? p.b2[n] : 0.f;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mrow = (r & 3) + 8 * (r >> 2) + 4 * hh;
      const int oy = ry0 + (mrow >> 4), ox = x0 + (mrow & 15);
      if (nok && oy < p.H && ox < p.W) {
        const float v = acc2[j][r] + bv;
        p.out[((long long)oy * p.W + ox) * p.ldo + n] = v;
        psum += v;
      }
    }
    psum += __shfl_xor(psum, 32);
    if (hh == 0) ps[wid * 192 + n] = psum;
  }
  __syncthreads();
  if (p.pool_part)
    for (int c = tid; c < 192; c += 512) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += ps[w * 192 + c];
      p.pool_part[(long long)blockIdx.x * 192 + c] = s;
    }
}

extern "C" long long ff_cab_fused_pool_rows(int H, int W) { return H > 0 && W > 0 ? (long long)((H + 15) / 16) * ((W + 15) / 16) : -1; }

extern "C" int ff_cab_fused(const float* in, int ldi, const void* w1_img, const float* b1, const void* w2_img, const float* b2, float* out,
                            int ldo, int H, int W, int Cin, int Cmid, int Cout, float* pool_partials, void* stream) {
  FF_CHECK_ARG(in && w1_img && b1 && w2_img && b2 && out, "ff_cab_fused: null pointer");
  FF_CHECK_ARG(H > 0 && W > 0 && Cin > 0 && Cin % 4 == 0 && Cmid > 0 && Cmid <= 64 && Cmid % 4 == 0 && Cout > 0 && Cout <= 192,
               "ff_cab_fused: needs Cin %% 4 == 0, Cmid <= 64, Cout <= 192 (HAT's CAB: 180 -> 60 -> 180)");
  FF_CHECK_ARG(ldi >= Cin && ldi % 4 == 0 && ldo >= Cout && (((uintptr_t)in) & 15) == 0 && (((uintptr_t)w1_img) & 15) == 0 && (((uintptr_t)w2_img) & 15) == 0,
               "ff_cab_fused: 16-byte aligned input rows and weight images required");
  FF_CHECK_ARG((long long)H * W * ldi < (1LL << 31) && in != out, "ff_cab_fused: tensor too large for 32-bit offsets / in-place not supported");
  CabParams p;
  p.in = in; p.w1 = (const unsigned char*)w1_img; p.b1 = b1; p.w2 = (const unsigned char*)w2_img; p.b2 = b2; p.out = out; p.pool_part = pool_partials;
  p.H = H; p.W = W; p.Cin = Cin; p.ldi = ldi; p.Cmid = Cmid; p.Cout = Cout; p.ldo = ldo; p.nchunk = (Cin + 63) / 64;
  p.tiles_x = (W + 15) / 16; p.tiles_y = (H + 15) / 16;
  constexpr size_t lds = (size_t)CB_XB + 2 * CB_W1SLOT + CB_CB;
  static_assert(lds <= 160 * 1024 && 2 * CB_W2SLOT <= CB_XB && 8 * 192 * 4 <= CB_XB, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cab_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { ff_set_error("ff_cab_fused: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return FF_ERR_LAUNCH; }
    attr_set = true;
  }
  hipLaunchKernelGGL(cab_fused_kernel, dim3((unsigned)(p.tiles_x * p.tiles_y)), dim3(512), lds, (hipStream_t)stream, p);
  FF_LAUNCH_CHECK("ff_cab_fused");
  return FF_OK;
}

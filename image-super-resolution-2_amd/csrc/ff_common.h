// Shared helpers for the FreqFusion HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FF_OK 0
#define FF_ERR_ARG 1
#define FF_ERR_LAUNCH 2

extern "C" const char* ff_last_error(void);
void ff_set_error(const char* fmt, ...);

#define FF_CHECK_ARG(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      ff_set_error(__VA_ARGS__);         \
      return FF_ERR_ARG;                 \
    }                                    \
  } while (0)

#define FF_LAUNCH_CHECK(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      ff_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return FF_ERR_LAUNCH;                                                \
    }                                                                      \
  } while (0)

// activation codes shared by every epilogue (include/ff_kernels.h FF_ACT_*)
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_LRELU = 3, ACT_SIGMOID = 4 };

__device__ __forceinline__ float ff_act(float v, int act) {
  switch (act) {
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_LRELU: return v > 0.f ? v : 0.01f * v;
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

// GELU with the Abramowitz-Stegun 7.1.26 erf (|error| <= 1.5e-7): ~14 VALU ops instead of ~55 for libm erff with its
// two divergent branches.  Used by the split-bf16 kernels (whose own error is ~1e-5); the f32 parity kernels keep erff.
__device__ __forceinline__ float ff_gelu_fast(float v) {
  const float ax = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  float pl = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  pl = __builtin_fmaf(pl, t, 1.421413741f);
  pl = __builtin_fmaf(pl, t, -0.284496736f);
  pl = __builtin_fmaf(pl, t, 0.254829592f);
  pl *= t;
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
  const float er = __builtin_copysignf(1.0f - pl * e, v);
  return 0.5f * v * (1.0f + er);
}

// Plain-bf16 kernels only: GELU as x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3) -- |error| <= 4.8e-4 against the erf form, a
// sixteenth of the bf16 rounding the value (or the product it enters) receives next.  Two values at a time: the packed fp32
// multiply / fma / add take both in one instruction; 5 full-rate + 2 transcendental operations per value instead of 14 + 2.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 ff_gelu_sig2(f32x2 x) {
  const float A = -2.3022081f, B = -0.10294324f;       // -2 sqrt(2/pi) log2(e) * {1, 0.044715}
  const f32x2 x2 = x * x;
  const f32x2 pz = x2 * B + A;
  const f32x2 z = x * pz;
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(z[0]); e[1] = __builtin_amdgcn_exp2f(z[1]);
  const f32x2 d = e + 1.0f;
  f32x2 r;
  r[0] = __builtin_amdgcn_rcpf(d[0]); r[1] = __builtin_amdgcn_rcpf(d[1]);
  return x * r;
}

__device__ __forceinline__ float ff_act_fast(float v, int act) { return act == ACT_GELU ? ff_gelu_fast(v) : ff_act(v, act); }

// compile-time activation (epilogues dispatch ONCE per kernel on the runtime code, not once per element:
// a per-element switch costs ~10 scalar+vector instructions and a branch for each of the 64-96 outputs of a lane)
template <int ACT, bool FAST>
__device__ __forceinline__ float ff_act_c(float v) {
  if (ACT == ACT_GELU) return FAST ? ff_gelu_fast(v) : 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
  if (ACT == ACT_LRELU) return v > 0.f ? v : 0.01f * v;
  if (ACT == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}
template <int V> struct ff_ic { static constexpr int value = V; };
#define FF_DISPATCH_ACT(act, fn)                 \
  switch (act) {                                 \
    case ACT_GELU: fn(ff_ic<ACT_GELU>{}); break;       \
    case ACT_RELU: fn(ff_ic<ACT_RELU>{}); break;       \
    case ACT_LRELU: fn(ff_ic<ACT_LRELU>{}); break;     \
    case ACT_SIGMOID: fn(ff_ic<ACT_SIGMOID>{}); break; \
    default: fn(ff_ic<ACT_NONE>{}); break;             \
  }

// A wave reads its 32 token rows (K <= 192 floats each) COALESCED -- four 256-byte row segments per load instruction --
// and redistributes them through a wave-private LDS patch (32 rows x 68 dwords: 4-bank row shift, conflict-free
// ds_read_b128) into the MFMA B-operand order: v[st][j] = x[row l31][16 st + 8 hh + j].  Per-lane strided row reads
// (two lanes per row, 32 B per instruction and row) reach only ~1.7 TB/s; this form streams at the HBM rate.
#define FF_XS_ROW 68
template <int NPASS>
__device__ __forceinline__ void ff_wave_rows_to_frags(const float* __restrict__ x, int ldx, long long tok0, long long M, int K,
                                                      float* xs, int lane, float (&v)[4 * NPASS][8]) {
  const int l31 = lane & 31, hh = lane >> 5;
  const int rr = lane >> 4, cq = (lane & 15) * 4;
  // every pass's loads are issued before the first one is consumed: one exposed HBM latency per wave instead of NPASS
  f32x4 t[NPASS][8];
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 4 * j + rr, c = 64 * pass + cq;
      const bool ok = tok0 + r < M && c < K;
      const f32x4 u = *reinterpret_cast<const f32x4*>(x + (ok ? (tok0 + r) * ldx + c : 0));
      t[pass][j] = ok ? u : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(xs + (4 * j + rr) * FF_XS_ROW + cq) = t[pass][j];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xs + l31 * FF_XS_ROW + 16 * s4 + 8 * hh);
      const f32x4 b = *reinterpret_cast<const f32x4*>(xs + l31 * FF_XS_ROW + 16 * s4 + 8 * hh + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[4 * pass + s4][e] = a[e]; v[4 * pass + s4][4 + e] = b[e]; }
    }
  }
}

static inline int ff_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Bijective XCD-aware block remap (guide section 5 T1): blocks with equal (bid % 8) share an XCD/L2,
// so give each XCD a contiguous run of logical tile ids.
__device__ __forceinline__ int ff_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// Error channel + tiny runtime queries of the C-ABI library.
#include "ff_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void ff_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ff_last_error(void) { return g_err; }

// 2: ff_token_linear gained the statistics side output; ff_win_attn_fused, ff_token_projmlp, ff_dwconv3x3_ln, the metric and
// executor entry points were added (round 2)
extern "C" int ff_abi_version(void) { return 6; }   // 6: bf16 intermediate rows (io_bf16 / out_bf16 / xn_bf16 arguments), ff_sgfn_tail, ff_ocab_attn, ff_pool_vec_mlp; 5: nterms argument of the fused token / halo / NAFNet kernels; 4: training side (csrc/train_ops.hip), ff_rfft2 / ff_irfft2
// was 3:   // 3: ff_conv2d_bf16s gained kmul + shuffle = 1; ff_naf_front, ff_conv3x3_small, ff_chan_qkv, executor streams

// Number of compute units of the current device (bench.py sizes its roofline report with it).
extern "C" int ff_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}

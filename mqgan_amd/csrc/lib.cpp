// libmqgan_hip.so: error channel + ABI version.
#include "common.h"

namespace mq {
thread_local char g_err[512] = "";
thread_local char g_kernel[128] = "";
}

// 2: mqConvDesc gained `precision`; mq_conv_wgrad_bias, mq_taps_*, mq_cast_bf16, mq_conv_*_h added
// 3: bf16-storage kernels (mq_h2_*, mq_conv_*_h2, mq_*_h); losses split into sums / final (global normalisers under data
//    parallelism, per-element masks); deterministic codebook EMA with its own workspace query
// 5: mq_multi_gather_args (arena fill table by value), mq_last_kernel, mq_l2_screen_* / mq_l2_argmin_screened
extern "C" int mq_abi_version(void) { return 7; }
extern "C" const char *mq_last_error(void) { return mq::g_err; }
extern "C" const char *mq_last_kernel(void) { return mq::g_kernel; }

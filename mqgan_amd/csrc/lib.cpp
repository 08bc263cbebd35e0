// libmqgan_hip.so: error channel + ABI version.
#include "common.h"

namespace mq {
thread_local char g_err[512] = "";
thread_local char g_kernel[128] = "";
}

// 2: mqConvDesc gained `precision`; mq_conv_wgrad_bias, mq_taps_*, mq_cast_bf16, mq_conv_*_h added
// 3: bf16-storage kernels (mq_h2_*, mq_conv_*_h2, mq_*_h); losses split into sums / final (global normalisers under data
//    parallelism, per-element masks); deterministic codebook EMA with its own workspace query
// 4: mq_conv_dgrad_gated / mq_conv_dgrad_h2_gated (the producer's LeakyReLU' / mask in the dgrad store), mq_wino_* (F(2x2, 3x3)),
//    mq_spectral_prep_bands, mq_multi_gather
// 5: mq_multi_gather_args (arena fill table by value), mq_last_kernel, mq_l2_screen_* / mq_l2_argmin_screened
// 6: mq_wino3_* (F(3x3, 3x3), the polyphase *_s2d / *_d2s transforms, mq_wino3_dy_fused), mq_wino1d_* (F(m, 3) along the rows,
//    mq_wino1d_tile)
// 7: mq_winon_* (F(4x4) / F(4x5, 3x3): tile passed as (mh, mw)), mq_winow_* (F(4,3) / F(4,5) / F(2,7) along the columns),
//    mq_winon_dy_fused / mq_winon_dy_fused_rows
// 8: mq_conv_plan / mq_conv_fprop_planned / mq_conv_backward_planned (the algorithm choice and the Winograd / polyphase / nested
//    sequences behind the ABI, csrc/plan.cpp), mq_set_launch_hook; mq_winon_* take the 3 x 3 tile as well - mq_wino3_* and the
//    F(2x2, 3x3) functions mq_wino_* are gone
// 9: mq_set_pass_hook (HIP-event timing of the transform passes, algorithmic bytes per reduced-form layer); the screened codebook
//    search is three launches with every bf16 score formed once (same entry points; any d % 8 == 0 - tuned for d <= 512, where a
//    block's x tile stays in L2; longer rows are tested, not tuned)
// 10: the product library exports only what mqgan_amd/ops.py calls (plus version / error / plan-configuration helpers): the transform
//    building blocks mq_winon_* / mq_wino1d_* / mq_winow_* and mq_winofused_fprop are internal (MQ_BLOCKS_API) and exported by the
//    test-variant build alone (make variants), which also carries the rejected kernel variants; the two hooks are named
//    mq_debug_set_*_hook; the screened codebook search's bound comes from measured rounding residuals
// 11: mq_comm_unique_id / mq_comm_init / mq_allreduce_bucket / mq_comm_destroy (the gradient-bucket all-reduce over RCCL, resolved
//    with dlopen at first use: csrc/collective.cpp)
extern "C" int mq_abi_version(void) { return 11; }
extern "C" const char *mq_last_error(void) { return mq::g_err; }
extern "C" const char *mq_last_kernel(void) { return mq::g_kernel; }

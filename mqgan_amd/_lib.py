"""ctypes binding of libmqgan_hip.so (C ABI: include/mqgan_hip.h).

There is no CPU fallback: if the library is missing, or a tensor is not a
contiguous fp32/int HIP tensor, the call raises.  PyTorch only supplies device
memory and the current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MQGAN_HIP_LIB: load another build of the same ABI (kernel experiments); the default is the in-tree library
LIB_PATH = os.environ.get("MQGAN_HIP_LIB") or os.path.join(_HERE, "libmqgan_hip.so")
_lib = None
ABI_VERSION = 11


class MqganHipError(RuntimeError):
    pass


class FsqDesc(C.Structure):
    _fields_ = [("d", C.c_int), ("levels", C.c_int * 8), ("half_width", C.c_int * 8), ("basis", C.c_int * 8),
                ("half_l", C.c_float * 8), ("offset", C.c_float * 8), ("shift", C.c_float * 8),
                ("thresh", (C.c_float * 8) * 8)]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "H", "W", "C", "K", "R", "S", "P", "Q", "stride_h", "stride_w", "pad_h",
                                       "pad_w", "dil_h", "dil_w", "groups", "precision")]


class PlanCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("wino_minc", "wino_minc_explicit", "wino_tile", "s2d_minc", "wino1d_minc", "wino1d_tile",
                                       "winow_minc", "keep_mib")]


class ConvPlan(C.Structure):
    _fields_ = [("form", C.c_int), ("mh", C.c_int), ("mw", C.c_int), ("nested", C.c_int), ("leaky", C.c_int), ("npos", C.c_int),
                ("tiles", C.c_int64), ("desc", ConvDesc), ("gemm", ConvDesc), ("inner", ConvDesc), ("keep_bytes", C.c_size_t),
                ("ws_fprop", C.c_size_t), ("ws_backward", C.c_size_t), ("flops", C.c_double), ("keep_filters_only", C.c_int),
                ("keep_uc_bytes", C.c_size_t)]


FORM_DIRECT, FORM_WINO2D, FORM_S2D, FORM_WINO1D, FORM_WINOW = range(5)
LaunchHook = C.CFUNCTYPE(None, C.c_int, C.c_double, C.c_char_p)
PassHook = C.CFUNCTYPE(None, C.c_int, C.c_double, C.c_char_p)


class Cb2dWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wdw", "bdw", "w1", "b1", "w2", "b2")]


class PointwiseDesc(C.Structure):
    _fields_ = [("rows", C.c_int64), ("C", C.c_int), ("act", C.c_int), ("mask_pre", C.c_int), ("mask_post", C.c_int),
                ("d1", C.c_int64), ("d2", C.c_int64), ("d3", C.c_int64), ("p_drop", C.c_float), ("seed", C.c_uint64)]


_P, _I, _L, _F, _D, _Z, _U64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t, C.c_uint64

# name -> (restype, argtypes); every symbol declared in include/mqgan_hip.h
SIGNATURES = {
    "mq_abi_version": (C.c_int, []),
    "mq_last_error": (C.c_char_p, []),
    "mq_last_kernel": (C.c_char_p, []),
    "mq_fsq_forward": (_I, [C.POINTER(FsqDesc), _P, _L, _P, _P, _P]),
    "mq_fsq_backward": (_I, [C.POINTER(FsqDesc), _P, _P, _L, _P, _P]),
    "mq_fsq_decode": (_I, [C.POINTER(FsqDesc), _P, _L, _P, _P]),
    "mq_l2_codebook_norms": (_I, [_P, _I, _I, _P, _P]),
    "mq_l2_argmin_workspace": (_Z, [_L]),
    "mq_l2_argmin": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _P, _P]),
    "mq_l2_screen_supported": (_I, [_L, _I, _I]),
    "mq_l2_screen_codebook_bytes": (_Z, [_I, _I]),
    "mq_l2_screen_workspace": (_Z, [_L, _I, _I]),
    "mq_l2_screen_prepare": (_I, [_P, _I, _I, _P, _P]),
    "mq_l2_argmin_screened": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _P, _Z, _P]),
    "mq_l2_screen_stats": (_I, [_P, _L, _I, _I, _P, _P, _P, _P]),
    "mq_codebook_ema_workspace": (_Z, [_L, _I, _I]),
    "mq_codebook_ema_update": (_I, [_P, _P, _L, _I, _I, _F, _F, _P, _P, _P, _P, _Z, _P]),
    "mq_conv_fprop": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "mq_conv_dgrad": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    "mq_conv_dgrad_gated": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _L, _L, _L, _F, _P, _P]),
    "mq_conv_fprop_leaky": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _L, _L, _L, _F, _P, _P]),
    "mq_cast_bf16": (_I, [_P, _P, _L, _P]),
    "mq_conv_fprop_h": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "mq_conv_dgrad_h": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    "mq_h2_supported": (_I, [C.POINTER(ConvDesc)]),
    "mq_h2_wgrad_supported": (_I, [C.POINTER(ConvDesc)]),
    "mq_h2_weight_elems": (_Z, [C.POINTER(ConvDesc), _I]),
    "mq_h2_weight_image": (_I, [C.POINTER(ConvDesc), _I, _P, _P, _P]),
    "mq_conv_fprop_h2": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _L, _L, _L, _I, _F, _I, _P, _P]),
    "mq_conv_dgrad_h2": (_I, [C.POINTER(ConvDesc), _P, _P, _I, _P, _P]),
    "mq_conv_dgrad_h2_gated": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _L, _L, _L, _F, _I, _P, _P]),
    "mq_conv_wgrad_h2_workspace": (_Z, [C.POINTER(ConvDesc)]),
    "mq_conv_wgrad_h2": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _Z, _P]),
    "mq_conv_wgrad_workspace": (_Z, [C.POINTER(ConvDesc)]),
    "mq_conv_wgrad": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _Z, _P]),
    "mq_conv_wgrad_bias_workspace": (_Z, [C.POINTER(ConvDesc)]),
    "mq_conv_wgrad_bias": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _Z, _P]),
    "mq_colsum": (_I, [_P, _L, _I, _P, _P, _P]),
    "mq_taps_gather": (_I, [C.POINTER(ConvDesc), _I, _P, _P, _P]),
    "mq_taps_scatter": (_I, [C.POINTER(ConvDesc), _I, _P, _P, _P]),
    "mq_taps_reduce": (_I, [C.POINTER(ConvDesc), _I, _P, _P, _P, _P]),
    "mq_taps_expand": (_I, [C.POINTER(ConvDesc), _I, _P, _P, _P]),
    "mq_weight_prep": (_I, [_I, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "mq_weight_prep_bwd": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "mq_spectral_prep": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P]),
    "mq_spectral_prep_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "mq_spectral_prep_bands": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "mq_spectral_prep_bands_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "mq_convblock2d_keep_bytes": (_Z, [_I, _I, _I]),
    "mq_convblock2d_fwd": (_I, [C.POINTER(Cb2dWeights), _P, _P, _I, _I, _I, _F, _U64, _P, _P, _P, _P]),
    "mq_convblock2d_bwd_workspace": (_Z, [_I, _I, _I]),
    "mq_convblock2d_bwd": (_I, [C.POINTER(Cb2dWeights), _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P,
                                _P, _P]),
    "mq_pointwise_fwd": (_I, [C.POINTER(PointwiseDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "mq_pointwise_bwd": (_I, [C.POINTER(PointwiseDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mq_pointwise_fwd_h": (_I, [C.POINTER(PointwiseDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "mq_pointwise_bwd_h": (_I, [C.POINTER(PointwiseDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mq_cast_f32": (_I, [_P, _P, _L, _P]),
    "mq_avgpool_t2_fwd_h": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "mq_avgpool_t2_bwd_h": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "mq_upcat_fwd_h": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mq_upcat_bwd_h": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "mq_sequence_mask": (_I, [_P, _I, _I, _P, _P]),
    "mq_mask_pool": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "mq_mask_upsample2": (_I, [_P, _I, _I, _P, _P]),
    "mq_avgpool_t2_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "mq_avgpool_t2_bwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "mq_upcat_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mq_upcat_bwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "mq_band_split_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "mq_band_split_bwd": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "mq_concat2": (_I, [_P, _P, _L, _I, _I, _P, _P]),
    "mq_axpby": (_I, [_P, _P, _F, _F, _L, _P, _P]),
    "mq_cam_pool": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "mq_sigmoid_add": (_I, [_P, _P, _L, _P, _P]),
    "mq_sigmoid_add_bwd": (_I, [_P, _P, _L, _P, _P]),
    "mq_relu_fwd": (_I, [_P, _L, _P, _P]),
    "mq_relu_bwd": (_I, [_P, _P, _L, _P, _P]),
    "mq_sam_pool": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "mq_sam_apply": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "mq_cbam_bwd_workspace": (_Z, [_I, _I, _I, _I]),
    "mq_cbam_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "mq_cam_pool_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "mq_se_squeeze": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "mq_se_scale_fwd": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "mq_se_scale_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "mq_se_squeeze_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "mq_mel_loss_sums": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "mq_mel_loss_final": (_I, [_P, _I, _I, _I, _P, _P]),
    "mq_mel_loss_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P]),
    "mq_lsgan_d_sums": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "mq_lsgan_ema": (_I, [_P, _I, _F, _P, _P, _P]),
    "mq_lsgan_d_lecam": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "mq_lsgan_d_final": (_I, [_P, _P, _P, _P]),
    "mq_lsgan_d_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "mq_lsgan_g_sums": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "mq_ratio_final": (_I, [_P, _F, _P, _P]),
    "mq_lsgan_g_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "mq_masked_mae_sums": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "mq_masked_mae_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _F, _P, _P]),
    "mq_wino1d_tile": (_I, [_I]),
    "mq_winow_tile": (_I, [_I]),
    "mq_plan_cfg_default": (None, [C.POINTER(PlanCfg)]),
    "mq_form_name": (C.c_char_p, [_I]),
    "mq_conv_plan": (_I, [C.POINTER(ConvDesc), _I, C.POINTER(PlanCfg), C.POINTER(ConvPlan)]),
    "mq_conv_fprop_planned": (_I, [C.POINTER(ConvPlan), _P, _P, _P, _I, _F, _P, _L, _L, _L, _P, _P, _P, _Z, _P]),
    "mq_conv_backward_planned": (_I, [C.POINTER(ConvPlan), _P, _P, _P, _P, _P, _P, _F, _P, _L, _L, _L, _P, _P, _P, _P, _Z, _P]),
    "mq_debug_set_launch_hook": (None, [LaunchHook]),
    "mq_debug_set_pass_hook": (None, [PassHook]),
    "mq_sumsq": (_I, [_P, _L, _P, _P, _P]),
    "mq_multi_gather": (_I, [_P, _I, _L, _P, _P]),
    "mq_multi_gather_args": (_I, [_P, _I, _P, _P]),
    "mq_adam_step": (_I, [_P, _P, _P, _P, _L, _D, _D, _D, _D, _I, _P, _F, _F, _P]),
    "mq_comm_unique_id": (_I, [_P]),
    "mq_comm_init": (_I, [C.POINTER(_P), _I, _I, _P]),
    "mq_allreduce_bucket": (_I, [_P, _P, _Z, _P]),
    "mq_comm_destroy": (_I, [_P]),
}

# Exported by the TEST-VARIANT build only (make -C mqgan_amd/csrc variants -> libmqgan_hip_variants.so): the transform building blocks
# behind the planned convolution calls and the fused-Winograd prototype (include/mqgan_hip.h: MQ_BLOCKS_API).  Nothing in this package
# calls them; tests/ and tools/ do, through load_variants().
VARIANT_SIGNATURES = {
    "mq_winon_input_transform": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "mq_winon_output_transform": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _P, _L, _L, _L, _P, _I, _I, _P]),
    "mq_winon_input_transform_s2d": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "mq_winon_weight_transform_s2d": (_I, [_P, _I, _I, _I, _P, _I, _I, _P]),
    "mq_winon_output_transform_d2s": (_I, [_P, _I, _I, _I, _I, _P, _F, _P, _L, _L, _L, _P, _I, _I, _P]),
    "mq_winon_wgrad_inverse_s2d": (_I, [_P, _I, _I, _P, _I, _I, _P]),
    "mq_winon_weight_transform": (_I, [_P, _I, _I, _I, _P, _I, _I, _P]),
    "mq_winon_wgrad_inverse": (_I, [_P, _I, _I, _P, _I, _I, _P]),
    "mq_winon_dy_fused_rows": (_I, [_I, _I, _I, _I, _I, _I]),
    "mq_winon_dy_fused": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P]),
    "mq_wino1d_input_transform": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "mq_wino1d_weight_transform": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mq_wino1d_output_transform": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _L, _L, _L, _P, _P]),
    "mq_wino1d_dy_rows": (_I, [_I, _I, _I, _I, _I, _I]),
    "mq_wino1d_dy_transform": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mq_wino1d_dx_transform": (_I, [_P, _I, _I, _I, _I, _I, _P, _F, _P, _L, _L, _L, _P, _P]),
    "mq_wino1d_wgrad_inverse": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "mq_winow_input_transform": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "mq_winow_weight_transform": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mq_winow_output_transform": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _L, _L, _L, _P, _P]),
    "mq_winow_dy_rows": (_I, [_I, _I, _I, _I, _I, _I]),
    "mq_winow_dy_transform": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mq_winow_dx_transform": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _F, _P, _L, _L, _L, _P, _P]),
    "mq_winow_wgrad_inverse": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "mq_winofused_fprop": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
}


def load():
    """Load the shared library (once) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MqganHipError(
            f"{LIB_PATH} not found: the HIP kernels are not built.  Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C mqgan_amd/csrc`).  mqgan_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.mq_abi_version() != ABI_VERSION:       # a stale build would misread mqConvDesc (struct layout changed)
        raise MqganHipError(f"{LIB_PATH} has ABI version {lib.mq_abi_version()}, this package needs {ABI_VERSION}: rebuild "
                            "(`make -C mqgan_amd/csrc`)")
    _lib = lib
    return lib


VARIANTS_PATH = os.path.join(_HERE, "libmqgan_hip_variants.so")
_variants = None


def load_variants():
    """The test-variant library (tests / tools / micro-benchmarks only): every entry point of the product library plus the
    building blocks of VARIANT_SIGNATURES, and the measured-and-rejected kernel variants behind their MQGAN_* switches."""
    global _variants
    if _variants is not None:
        return _variants
    if not os.path.exists(VARIANTS_PATH):
        raise MqganHipError(f"{VARIANTS_PATH} not found: build it with `make -C mqgan_amd/csrc variants` "
                            "(python -c 'import __graft_entry__ as g; g.build_variants()')")
    lib = C.CDLL(VARIANTS_PATH)
    for name, (res, args) in {**SIGNATURES, **VARIANT_SIGNATURES}.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.mq_abi_version() != ABI_VERSION:
        raise MqganHipError(f"{VARIANTS_PATH} has ABI version {lib.mq_abi_version()}, this package needs {ABI_VERSION}: rebuild")
    _variants = lib
    return lib


class use_variants:
    """Context manager for tests: every call of the package goes to the test-variant library inside the block (the same process
    keeps the product library for everything else)."""

    def __enter__(self):
        global _lib
        self._keep = load()
        _lib = load_variants()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._keep
        return False


def check(rc, what=""):
    if rc != 0:
        msg = load().mq_last_error().decode("utf-8", "replace")
        raise MqganHipError(f"{what}: {msg}" if what else msg)


def stream():
    """HIP stream every kernel of this call is launched on: the current stream of the CURRENT device.  ptr() refuses
    tensors that live on another device, so a launch can never pair device-N pointers with device-0's stream."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None, allow_none=False):
    """Device pointer of a contiguous HIP tensor (validated)."""
    if t is None:
        if allow_none:
            return C.c_void_p(0)
        raise MqganHipError("required tensor is None")
    if not t.is_cuda:
        raise MqganHipError("mqgan_amd ops need HIP device tensors (got a CPU tensor); there is no CPU path")
    if t.device.index != torch.cuda.current_device():
        raise MqganHipError(f"tensor lives on {t.device} but the current HIP device is cuda:{torch.cuda.current_device()}: "
                            "call torch.cuda.set_device(...) (Trainer / ScriptedPreEncoder do) before launching kernels")
    if not t.is_contiguous():
        raise MqganHipError(f"tensor of shape {tuple(t.shape)} is not contiguous")
    if dtype is not None and t.dtype != dtype:
        raise MqganHipError(f"expected dtype {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def fptr(t, allow_none=False):
    return ptr(t, torch.float32, allow_none)

"""ctypes binding of the C ABI declared in include/frcnn_hip.h (libfrcnn_hip.so).

The library is built in-tree by csrc/build.py (hipcc --offload-arch=gfx950).  There is no CPU
fallback: if the shared object is missing, or no MI355X is visible, `load()` raises.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfrcnn_hip.so")

_P = ctypes.c_void_p
_I = ctypes.c_int
_D = ctypes.c_double
_F = ctypes.c_float
_S = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/frcnn_hip.h one to one
SIGNATURES = {
    "frcnn_abi_version": (_I, []),
    "frcnn_device_count": (_I, []),
    "frcnn_set_tuning": (_I, [ctypes.c_char_p, ctypes.c_char_p]),
    "frcnn_get_tuning": (_I, [ctypes.c_char_p, ctypes.c_char_p, _I]),
    "frcnn_reset_tuning": (_I, []),
    "frcnn_nms_workspace_bytes": (_S, [_I]),
    "frcnn_nms": (_I, [_P, _I, _D, _I, _P, _P, _P, _S, _P]),
    "frcnn_nms_batched_workspace_bytes": (_S, [_I, _I]),
    "frcnn_nms_batched": (_I, [_P, _I, _I, _D, _I, _P, _P, _P, _S, _P]),
    "frcnn_proposals_workspace_bytes": (_S, [_I, _I, _I, _I]),
    "frcnn_proposals": (_I, [_P, _P, _I, _I, _I, _P, _I, _I, _I, _F, _I, _I, _D, _P, _P, _P, _P, _P, _S, _P]),
    "frcnn_roi_pool_workspace_bytes": (_S, [_I, _I, _I]),
    "frcnn_chw_to_hwc": (_I, [_P, _I, _I, _I, _P, _P]),
    "frcnn_roi_pool_fwd_hwc": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _F, _P, _P, _P]),
    "frcnn_roi_pool_fwd_chw": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _F, _P, _P, _P, _S, _P]),
    "frcnn_roi_pool_fwd_chw_bf16": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _F, _P, _P]),
    "frcnn_roi_pool_fwd_blk_bf16": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _F, _P, _I, _P]),
    "frcnn_roi_pool_fwd_chw_f32s": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _F, _P, _P]),
    "frcnn_roi_pool_fwd": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _F, _P, _P, _P, _S, _P]),
    "frcnn_roi_pool_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "frcnn_pack_conv3x3_w": (_I, [_P, _I, _I, _P, _P]),
    "frcnn_conv3x3_workspace_bytes": (_S, [_I, _I, _I, _I]),
    "frcnn_conv3x3_workspace_init": (_I, [_P, _S, _P]),
    "frcnn_conv3x3_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_conv3x3_f32_cfg": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_maxpool2x2_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "frcnn_rpn_heads_padded_channels": (_I, [_I]),
    "frcnn_rpn_heads_pack": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "frcnn_rpn_heads_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "frcnn_linear_workspace_bytes": (_S, [_I, _I, _I]),
    "frcnn_linear_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_head_decode": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "frcnn_head_decode_stacked": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "frcnn_preprocess_u8": (_I, [_P, _I, _I, _I, _P, _D, _I, _I, _P, _P]),
    "frcnn_class_dets": (_I, [_P, _P, _I, _I, _P, _P]),
    "frcnn_bbox_transform_inv": (_I, [_P, _P, _I, _I, _P, _P]),
    "frcnn_clip_boxes": (_I, [_P, _I, _I, _I, _P]),
    "frcnn_softmax_rows": (_I, [_P, _I, _I, _P, _P]),
    "frcnn_conv_f32_ex": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_f32s_pack_conv_w": (_I, [_P, _I, _I, _P, _P]),
    "frcnn_f32s_from_nchw_f32": (_I, [_P, _I, _I, _I, _P, _P]),
    "frcnn_f32s_to_nchw_f32": (_I, [_P, _I, _I, _I, _P, _P]),
    "frcnn_conv3x3_f32s": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "frcnn_conv1_f32s": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "frcnn_conv1_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "frcnn_conv1_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "frcnn_f32s_pack_from_packed": (_I, [_P, _I, _I, _I, _P, _P]),
    "frcnn_f32s_pack_many": (_I, [_P, _I, _P]),
    "frcnn_conv_wgrad_f32s": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_conv1_f32s_train": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "frcnn_conv3x3_f32s_train": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_f32s_split": (_I, [_P, _S, _P, _P]),
    "frcnn_f32s_join": (_I, [_P, _S, _P, _P]),
    "frcnn_linear_f32s_workspace_bytes": (_S, [_I, _I, _I]),
    "frcnn_linear_f32s": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_conv_f32s_workspace_bytes": (_S, [_I, _I, _I, _I]),
    "frcnn_conv_f32s_workspace_init": (_I, [_P, _S, _P]),
    "frcnn_conv3x3_f32s_ws": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_bf16_padded_channels": (_I, [_I]),
    "frcnn_bf16_pack_conv_w": (_I, [_P, _I, _I, _I, _P, _P]),
    "frcnn_bf16_from_nchw_f32": (_I, [_P, _I, _I, _I, _P, _P]),
    "frcnn_bf16_to_nchw_f32": (_I, [_P, _I, _I, _I, _P, _P]),
    "frcnn_conv_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "frcnn_conv_bf16_workspace_bytes": (_S, [_I, _I, _I, _I]),
    "frcnn_conv_bf16_plan": (_I, [_I, _I, _I, _I, _I, _I]),
    "frcnn_conv_bf16_workspace_init": (_I, [_P, _S, _P]),
    "frcnn_conv_bf16_ws": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_maxpool2x2_bf16": (_I, [_P, _P, _I, _I, _I, _P]),
    "frcnn_conv1_pair_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "frcnn_f32_to_bf16": (_I, [_P, _S, _P, _P]),
    "frcnn_linear_bf16_workspace_bytes": (_S, [_I, _I, _I]),
    "frcnn_linear_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_linear_bf16_tiled_bytes": (_S, [_I, _I]),
    "frcnn_linear_bf16_tile_w": (_I, [_P, _I, _I, _P, _P]),
    "frcnn_linear_bf16_tiled_workspace_bytes": (_S, [_I, _I, _I]),
    "frcnn_linear_bf16_tiled": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_softmax_channels_f32": (_I, [_P, _I, _I, _P, _P]),
    "frcnn_rpn_heads_bf16": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "frcnn_roi_pool_fwd_chw_f16": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _F, _P, _P]),
    "frcnn_roi_pool_fwd_blk_f16": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _F, _P, _I, _P]),
    # the fp16 twins of the 16-bit chain (csrc/conv_f16.hip ...): same signatures
    "frcnn_f16_to_nchw_f32": (_I, [_P, _I, _I, _I, _P, _P]),
    "frcnn_f16_padded_channels": (_I, [_I]),
    "frcnn_f16_pack_conv_w": (_I, [_P, _I, _I, _I, _P, _P]),
    "frcnn_f16_from_nchw_f32": (_I, [_P, _I, _I, _I, _P, _P]),
    "frcnn_conv_f16_workspace_bytes": (_S, [_I, _I, _I, _I]),
    "frcnn_conv_f16_workspace_init": (_I, [_P, _S, _P]),
    "frcnn_conv_f16_ws": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_rpn_heads_f16": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "frcnn_conv_f16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "frcnn_conv_f16_plan": (_I, [_I, _I, _I, _I, _I, _I]),
    "frcnn_maxpool2x2_f16": (_I, [_P, _P, _I, _I, _I, _P]),
    "frcnn_f32_to_f16": (_I, [_P, _S, _P, _P]),
    "frcnn_linear_f16_workspace_bytes": (_S, [_I, _I, _I]),
    "frcnn_linear_f16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_conv1_pair_f16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "frcnn_linear_f16_tiled_bytes": (_S, [_I, _I]),
    "frcnn_linear_f16_tile_w": (_I, [_P, _I, _I, _P, _P]),
    "frcnn_linear_f16_tiled_workspace_bytes": (_S, [_I, _I, _I]),
    "frcnn_linear_f16_tiled": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_im2col7x7s2_f32": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "frcnn_maxpool3x3s2_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "frcnn_subsample2_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "frcnn_bbox_overlaps_f64": (_I, [_P, _I, _P, _I, _P, _P]),
    "frcnn_anchor_target_workspace_bytes": (_S, [_I, _I, _I, _I]),
    "frcnn_anchor_target": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _S, _P]),
    "frcnn_rpn_loss": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P]),
    "frcnn_rcnn_loss": (_I, [_P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P]),
    "frcnn_mul_f32": (_I, [_P, _P, _S, _P, _P]),
    "frcnn_add_f32": (_I, [_P, _P, _S, _P, _P]),
    "frcnn_dropout_f32": (_I, [_P, _S, _F, ctypes.c_ulonglong, _P, _P, _P]),
    "frcnn_relu_bwd_f32": (_I, [_P, _P, _S, _P]),
    "frcnn_gather_rows_f32": (_I, [_P, _P, _I, _I, _P, _P]),
    "frcnn_scatter_rows_f32": (_I, [_P, _P, _I, _I, _P, _I, _P]),
    "frcnn_maxpool2x2_bwd_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "frcnn_maxpool2x2_bwd_idx_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "frcnn_conv_dgrad_unpool_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_bias_grad_workspace_bytes": (_S, [_I, _I]),
    "frcnn_bias_grad_f32": (_I, [_P, _I, _I, _P, _P, _S, _P]),
    "frcnn_pack_conv_dgrad_w": (_I, [_P, _I, _I, _I, _P, _P]),
    "frcnn_pack_conv_dgrad_w_many": (_I, [_P, _I, _P]),
    "frcnn_conv_wgrad_workspace_bytes": (_S, [_I, _I, _I, _I, _I]),
    "frcnn_conv_wgrad_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _S, _P]),
    "frcnn_sgd_momentum_wd": (_I, [_P, _P, _P, _S, _F, _F, _F, _P]),
    "frcnn_transpose_f32": (_I, [_P, _I, _I, _P, _P]),
}


class FrcnnError(RuntimeError):
    pass


ERR_UNSUPPORTED = -2            # include/frcnn_hip.h: valid arguments, but this entry's kernel form declines the shape (the header names the detour)


def check(status, what):
    if status != 0:
        if status == -1:
            raise ValueError("%s: invalid argument (FRCNN_ERR_INVALID)" % what)
        if status == ERR_UNSUPPORTED:
            raise ValueError("%s: shape not covered by this kernel form (FRCNN_ERR_UNSUPPORTED)" % what)
        raise FrcnnError("%s failed: hipError_t %d" % (what, -status - 1000))


def bind(path):
    """dlopen `path` and attach the prototypes of every entry point the header declares."""
    if not os.path.exists(path):
        raise FrcnnError("HIP library not built: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    # the reference's own C FFI (models/gpu_nms.hpp:9-10): void _nms(int*, int*, const float*, int, int, float, int)
    lib._nms.restype = None
    lib._nms.argtypes = [_P, ctypes.POINTER(ctypes.c_int), _P, _I, _I, _F, _I]
    from . import tuning
    tuning.register(lib)
    return lib


_lib = None


def load():
    """The product library.  Raises when it is missing or when no GPU is visible -- never falls back."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm ships its own libamdhip64.so.7 (same soname as /opt/rocm's).  The process must hold ONE
        # HIP runtime -- the one torch allocates from -- so make torch load and initialise its copy first;
        # libfrcnn_hip.so's NEEDED entry then resolves to that already-loaded runtime.  (The other order
        # leaves torch bound to a runtime it was not built for and torch.cuda reports no device.)
        import torch
        if not torch.cuda.is_available():
            raise FrcnnError("no HIP device visible to PyTorch: the MI355X path cannot run (no CPU fallback)")
        lib = bind(LIB_PATH)
        n = lib.frcnn_device_count()
        if n <= 0:
            raise FrcnnError("libfrcnn_hip.so loaded but no HIP device is visible (frcnn_device_count=%d)" % n)
        _lib = lib
    return _lib

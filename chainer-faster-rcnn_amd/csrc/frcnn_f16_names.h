// frcnn_f16_names.h -- the entry points of the 16-bit chain under their fp16 names: included (before anything else) by the four fp16 translation units
// conv_f16.hip, conv_f16_pair.hip, linear_f16.hip, roi_f16.hip, which then include the bf16 sources unchanged with FRCNN_HALF_F16 defined (frcnn_intrin.h: pack / widen / MFMA
// become the fp16 instructions).  Every declaration include/frcnn_hip.h makes for a name below is thereby also the declaration of its fp16 twin: same signature,
// same layouts, same workspaces, same error codes.  Longest names first is not needed: the preprocessor matches whole identifiers.
#pragma once
#ifndef FRCNN_HALF_F16
#error "frcnn_f16_names.h is for the fp16 translation units only"
#endif
#define frcnn_bf16_to_nchw_f32 frcnn_f16_to_nchw_f32
#define frcnn_bf16_padded_channels frcnn_f16_padded_channels
#define frcnn_bf16_pack_conv_w frcnn_f16_pack_conv_w
#define frcnn_bf16_from_nchw_f32 frcnn_f16_from_nchw_f32
#define frcnn_conv_bf16_workspace_bytes frcnn_conv_f16_workspace_bytes
#define frcnn_conv_bf16_workspace_init frcnn_conv_f16_workspace_init
#define frcnn_conv_bf16_ws frcnn_conv_f16_ws
#define frcnn_rpn_heads_bf16 frcnn_rpn_heads_f16
#define frcnn_conv_bf16 frcnn_conv_f16
#define frcnn_conv_bf16_plan frcnn_conv_f16_plan
#define frcnn_maxpool2x2_bf16 frcnn_maxpool2x2_f16
#define frcnn_f32_to_bf16 frcnn_f32_to_f16
#define frcnn_linear_bf16_workspace_bytes frcnn_linear_f16_workspace_bytes
#define frcnn_linear_bf16 frcnn_linear_f16
#define frcnn_conv1_pair_bf16 frcnn_conv1_pair_f16
#define frcnn_linear_bf16_tiled_bytes frcnn_linear_f16_tiled_bytes
#define frcnn_linear_bf16_tile_w frcnn_linear_f16_tile_w
#define frcnn_linear_bf16_tiled_workspace_bytes frcnn_linear_f16_tiled_workspace_bytes
#define frcnn_linear_bf16_tiled frcnn_linear_f16_tiled
#define frcnn_roi_pool_fwd_chw_bf16 frcnn_roi_pool_fwd_chw_f16
#define frcnn_roi_pool_fwd_blk_bf16 frcnn_roi_pool_fwd_blk_f16

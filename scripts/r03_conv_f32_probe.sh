#!/bin/bash
# fp32 forward convolution: 16-byte halo DMA (this tree) against the 4-byte form (scripts/micro/_bin/libfrcnn_hip_base.so), then correctness on the GPU
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r03wg; mkdir -p $O; B=scripts/micro/_bin
{ echo "== base (4-byte halo pieces)"; $B/conv_f32_micro_base; echo "== this tree (16-byte halo pieces)"; $B/conv_f32_micro; echo "== base again"; $B/conv_f32_micro_base conv3_2 conv5_1; } > $O/conv_f32_micro.txt 2>&1
cat $O/conv_f32_micro.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 -k "conv or fp32 or forward or end_to_end or train" > $O/pytest_conv.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_conv.log

#!/bin/bash
# Round 2: timing ablations of conv_f32s_kernel (tuning build: FRCNN_TIMING_ABLATIONS=1 -- rebuilt on the box, never shipped)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02l}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp chainer-faster-rcnn_amd/libfrcnn_hip.so /tmp/lib_keep.so
FRCNN_TIMING_ABLATIONS=1 python -c "
import importlib.util,sys
spec=importlib.util.spec_from_file_location('b','chainer-faster-rcnn_amd/csrc/build.py'); m=importlib.util.module_from_spec(spec); spec.loader.exec_module(m); print(m.build(force=True))" > $O/build.log 2>&1; tail -1 $O/build.log
ABLS=1,4,5,8,9 timeout 600 python scripts/conv_f32s_bench.py conv1_2 conv3_2 conv4_2 conv5_1 > $O/f32s_abl.log 2>&1; grep -v amdgpu.ids $O/f32s_abl.log | tail -30
cp /tmp/lib_keep.so chainer-faster-rcnn_amd/libfrcnn_hip.so

#!/bin/bash
# builds the standalone micro-benchmarks (no torch) into scripts/micro/_bin (git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/../.."
mkdir -p scripts/micro/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/store_micro.hip -o scripts/micro/_bin/store_micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip scripts/micro/roi_micro.cpp -I include -L chainer-faster-rcnn_amd -lfrcnn_hip -Wl,-rpath,'$ORIGIN/../../../chainer-faster-rcnn_amd' -o scripts/micro/_bin/roi_micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip scripts/micro/conv_bf16_micro.cpp -I include -L chainer-faster-rcnn_amd -lfrcnn_hip -Wl,-rpath,'$ORIGIN/../../../chainer-faster-rcnn_amd' -o scripts/micro/_bin/conv_bf16_micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip scripts/micro/conv_pair_micro.cpp -I include -L chainer-faster-rcnn_amd -lfrcnn_hip -Wl,-rpath,'$ORIGIN/../../../chainer-faster-rcnn_amd' -o scripts/micro/_bin/conv_pair_micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip scripts/micro/wgrad_micro.cpp -I include -L chainer-faster-rcnn_amd -lfrcnn_hip -Wl,-rpath,'$ORIGIN/../../../chainer-faster-rcnn_amd' -o scripts/micro/_bin/wgrad_micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip scripts/micro/conv_f32_micro.cpp -I include -L chainer-faster-rcnn_amd -lfrcnn_hip -Wl,-rpath,'$ORIGIN/../../../chainer-faster-rcnn_amd' -o scripts/micro/_bin/conv_f32_micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-inline-asm -Wno-unused-value scripts/micro/mfma_dma_micro.hip -o scripts/micro/_bin/mfma_dma_micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-inline-asm -Wno-unused-value scripts/micro/dma_align_micro.hip -o scripts/micro/_bin/dma_align_micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_peak_micro.hip -o scripts/micro/_bin/mfma_peak_micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip scripts/micro/linear_bf16_micro.cpp -I include -L chainer-faster-rcnn_amd -lfrcnn_hip -Wl,-rpath,'$ORIGIN/../../../chainer-faster-rcnn_amd' -o scripts/micro/_bin/linear_bf16_micro
# timing-ablation build of roi_pool.hip alone (FRCNN_ROI_DBG is honoured; WRONG results by design) + the same harness against it
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DFRCNN_TIMING_ABLATIONS -I include -I chainer-faster-rcnn_amd/csrc -shared chainer-faster-rcnn_amd/csrc/roi_pool.hip chainer-faster-rcnn_amd/csrc/abi.hip -o scripts/micro/_bin/libroi_abl.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip scripts/micro/roi_micro.cpp -I include -L scripts/micro/_bin -lroi_abl -Wl,-rpath,'$ORIGIN' -o scripts/micro/_bin/roi_micro_abl

# timing-ablation build of linear_bf16.hip alone (FRCNN_LINEAR_RING_ABL is honoured; WRONG results by design) + the same harness against it ("new" only)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DFRCNN_TIMING_ABLATIONS -I include -I chainer-faster-rcnn_amd/csrc -shared chainer-faster-rcnn_amd/csrc/linear_bf16.hip chainer-faster-rcnn_amd/csrc/abi.hip -o scripts/micro/_bin/liblinear_abl.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -DLINEAR_MICRO_NEW_ONLY scripts/micro/linear_bf16_micro.cpp -I include -L scripts/micro/_bin -llinear_abl -Wl,-rpath,'$ORIGIN' -o scripts/micro/_bin/linear_bf16_micro_abl

# full research build of the library for the conv micro-benchmarks: opt-in.  -DFRCNN_TUNING_FORMS adds the measured-and-not-adopted kernel forms the product
# library no longer carries; -DFRCNN_TIMING_ABLATIONS the timing ablations (WRONG results by design)
if [ "${MICRO_ABL_LIB:-0}" = "1" ]; then
  mkdir -p scripts/micro/_bin/abl_obj
  for f in chainer-faster-rcnn_amd/csrc/*.hip; do
    o=scripts/micro/_bin/abl_obj/$(basename $f .hip).o
    if [ ! -f $o ] || [ $f -nt $o ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DFRCNN_TIMING_ABLATIONS -DFRCNN_TUNING_FORMS -I include -I chainer-faster-rcnn_amd/csrc -c $f -o $o & fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/micro/_bin/libfrcnn_hip_abl.so scripts/micro/_bin/abl_obj/*.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip scripts/micro/conv_bf16_micro.cpp -I include -L scripts/micro/_bin -lfrcnn_hip_abl -Wl,-rpath,'$ORIGIN' -o scripts/micro/_bin/conv_bf16_micro_abl
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip scripts/micro/wgrad_micro.cpp -I include -L scripts/micro/_bin -lfrcnn_hip_abl -Wl,-rpath,'$ORIGIN' -o scripts/micro/_bin/wgrad_micro_abl
  echo built-abl
fi
echo built
# research build of the fp32 convolution alone (-DFRCNN_SWEEP_FORMS: the round-6 sweep candidates 40..48 of frcnn_conv3x3_f32_cfg and the 1x1 tile forms of FRCNN_CONV1X1_CFG) + the same harness against it: opt-in
if [ "${MICRO_CONV_F32_FORMS:-0}" = "1" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DFRCNN_TUNING_FORMS -DFRCNN_SWEEP_FORMS -I include -I chainer-faster-rcnn_amd/csrc -shared chainer-faster-rcnn_amd/csrc/conv.hip chainer-faster-rcnn_amd/csrc/conv_f32s.hip chainer-faster-rcnn_amd/csrc/abi.hip -o scripts/micro/_bin/libconv_forms.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip scripts/micro/conv_f32_micro.cpp -I include -L scripts/micro/_bin -lconv_forms -Wl,-rpath,'$ORIGIN' -o scripts/micro/_bin/conv_f32_micro_forms
  echo built-conv-forms
fi

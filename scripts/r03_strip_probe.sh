#!/bin/bash
# strip-form probe (torch-free, seconds): variants vs the default pick with a bit comparison, then the timing ablations of each form
# (WRONG results by design: conv_bf16_micro_abl links scripts/micro/_bin/libfrcnn_hip_abl.so; MICRO_ABL_LIB=1 scripts/micro/build_micro.sh)
mkdir -p gpurun_out
B=scripts/micro/_bin
{
timeout 40 $B/conv_bf16_micro --check --modes "def 901 904 905 902 906" conv3_2 conv4_2
timeout 20 $B/conv_bf16_micro --check --modes "def 903 902" conv5_1 conv4_1
echo "--- ablations: 90<form><abl>, abl 1 no DMA, 2 no fragment reads, 3 both, 4 no MFMA, 8 no hand-over"
timeout 30 $B/conv_bf16_micro_abl --modes "9010 9011 9012 9013 9014 9018" conv3_2
timeout 30 $B/conv_bf16_micro_abl --modes "9020 9021 9022 9023 9024 9028" conv4_2
timeout 30 $B/conv_bf16_micro_abl --modes "9030 9031 9032 9033 9034 9038" conv5_1
FRCNN_BF16_STRIP_ABL=11 timeout 30 $B/conv_bf16_micro_abl --modes "9010" conv3_2
} > gpurun_out/strip_probe.txt 2>&1
cat gpurun_out/strip_probe.txt

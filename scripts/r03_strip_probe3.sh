#!/bin/bash
mkdir -p gpurun_out
B=scripts/micro/_bin
{
timeout 30 $B/mfma_peak_micro 1 20000
echo "--- all-zero operands (CONV_MICRO_ZERO=1) vs random: default pick, strip forms, and the strip form's MFMA-only ablation"
CONV_MICRO_ZERO=1 timeout 30 $B/conv_bf16_micro --modes "def 901" conv3_2
CONV_MICRO_ZERO=1 timeout 30 $B/conv_bf16_micro --modes "def 903" conv5_1
FRCNN_BF16_STRIP_ABL=11 CONV_MICRO_ZERO=1 timeout 30 $B/conv_bf16_micro_abl --modes "9010" conv3_2
FRCNN_BF16_STRIP_ABL=11 timeout 30 $B/conv_bf16_micro_abl --modes "9010" conv3_2
} > gpurun_out/strip_probe3.txt 2>&1
cat gpurun_out/strip_probe3.txt

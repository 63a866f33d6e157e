#!/bin/bash
# pitch-40 / immediate-offset / one-wait-per-group variants (911..913) against 901..903 (which now carry the M0-add DMA form and the early bias fetch), bit comparison with the FRCNN_BF16_STRIP=0 picks
mkdir -p gpurun_out
B=scripts/micro/_bin
{
FRCNN_BF16_STRIP=0 timeout 40 $B/conv_bf16_micro --check --modes "def 901 911" conv3_2 conv3_3
FRCNN_BF16_STRIP=0 timeout 40 $B/conv_bf16_micro --check --modes "def 902 912" conv4_1 conv4_2 conv4_3
FRCNN_BF16_STRIP=0 timeout 40 $B/conv_bf16_micro --check --modes "def 903 913" conv5_1
} > gpurun_out/strip_probe5.txt 2>&1
cat gpurun_out/strip_probe5.txt

#!/bin/bash
mkdir -p gpurun_out
B=scripts/micro/_bin
{
timeout 20 $B/mfma_peak_micro 1 20000
timeout 20 $B/mfma_peak_micro 2 20000
timeout 30 $B/conv_bf16_micro --check --modes "def 902 907" conv4_1 conv4_2 conv4_3
} > gpurun_out/strip_probe2.txt 2>&1
cat gpurun_out/strip_probe2.txt

#!/bin/bash
# the chain with the new default picks (strip forms where one round covers the launch) against the r03 picks (FRCNN_BF16_STRIP=0), bit comparison included
mkdir -p gpurun_out
B=scripts/micro/_bin
{
echo "=== default picks (strip rule on)"
timeout 40 $B/conv_bf16_micro --check --modes "def"
echo "=== FRCNN_BF16_STRIP=0 (the r03 picks)"
FRCNN_BF16_STRIP=0 timeout 40 $B/conv_bf16_micro
} > gpurun_out/strip_probe4.txt 2>&1
cat gpurun_out/strip_probe4.txt

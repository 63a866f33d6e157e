#!/bin/bash
# 909 (64 couts x 10 rows, two workgroups per CU) against the one-round default picks (forms A / B / C) on the layers those cover
mkdir -p gpurun_out
B=scripts/micro/_bin
{ timeout 60 $B/conv_bf16_micro --check --modes "def 909 908" conv3_2 conv3_3 conv4_1 conv4_2 conv4_3 conv5_1; } > gpurun_out/strip_probe7.txt 2>&1
cat gpurun_out/strip_probe7.txt

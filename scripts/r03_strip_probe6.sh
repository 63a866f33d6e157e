#!/bin/bash
# two-workgroups-per-CU strip variants (908: 64 couts x 12 rows, 909: 64 couts x 10 rows) on the launches of several rounds
mkdir -p gpurun_out
B=scripts/micro/_bin
{ timeout 60 $B/conv_bf16_micro --check --modes "def 908 909" conv1_2 conv2_1 conv2_2 conv3_1; } > gpurun_out/strip_probe6.txt 2>&1
cat gpurun_out/strip_probe6.txt

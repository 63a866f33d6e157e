#!/bin/bash
# the chain under the final default picks (strip form D / C rule) against conv_dma_bf16_kernel's picks ("old"), outputs compared bit for bit
mkdir -p gpurun_out
B=scripts/micro/_bin
{ timeout 60 $B/conv_bf16_micro --check --modes "def old"; } > gpurun_out/strip_probe8.txt 2>&1
cat gpurun_out/strip_probe8.txt

#!/bin/bash
# form D with the direct epilogue (910) against the default pick (D through the LDS transpose) on un-pooled layers
mkdir -p gpurun_out
{ timeout 8 scripts/micro/_bin/conv_bf16_micro --check --modes "def 910" conv2_1 conv3_1 conv3_2 conv4_2; } > gpurun_out/strip_probe9.txt 2>&1
cat gpurun_out/strip_probe9.txt

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r04e; mkdir -p $O; B=scripts/micro/_bin
{ timeout 200 $B/conv_bf16_micro --check --modes "def 911 901" conv2_2 conv3_1 conv3_2 conv3_3 conv4_1 conv4_2; } > $O/r04_conv_formE_micro_${TAG:-a}.txt 2>&1; cat $O/r04_conv_formE_micro_${TAG:-a}.txt

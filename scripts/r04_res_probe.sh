#!/bin/bash
# Round 4: the resident-weights producer / consumer forms (csrc/conv_bf16_res.h) on the four layers their rule picks, against conv_dma_bf16_kernel's and the
# strip forms' times, word by word (--check compares every mode with the default pick = the resident form)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r04r; mkdir -p $O; B=scripts/micro/_bin
{ for pr in 0 1 2; do echo "== FRCNN_BF16_RES_PRIO=$pr (0 none, 1 consumers first, 2 producers first)";
    FRCNN_BF16_RES_PRIO=$pr timeout 120 $B/conv_bf16_micro --check --modes "def old 910" conv1_2 conv2_1 conv2_2 conv3_1; done; } > $O/r04_conv_res_micro_${TAG:-a}.txt 2>&1; cat $O/r04_conv_res_micro_${TAG:-a}.txt

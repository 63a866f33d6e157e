#!/bin/bash
# Round 5: what a conv launch pays for COLD operands -- every launch of the timed graph reads its weights (w) / its input (x) from a different copy, more copies than the
# Infinity Cache holds -- on the 38 x 63 layers (strip form C), conv4_2 (form D) and conv2_1.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05l; mkdir -p $O
{ for c in "" w x wx; do echo "== CONV_MICRO_COLD=$c"; CONV_MICRO_COLD=$c CONV_MICRO_BURST=5 timeout 120 ./scripts/micro/_bin/conv_bf16_micro conv5_1 conv4_2 conv2_1; done; } > $O/r05_conv_cold_micro.txt 2>&1; cat $O/r05_conv_cold_micro.txt

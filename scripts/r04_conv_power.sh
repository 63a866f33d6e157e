#!/bin/bash
# Round 4: is the bf16 convolution itself running into the package power limit?  conv3_2 (the default pick: strip form D) back to back for ~5 s on random
# operands, then on all-zero operands, rocm-smi polled every ~0.15 s beside it.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r04w; mkdir -p $O; B=scripts/micro/_bin
( for i in $(seq 1 110); do echo "t=$(date +%s.%N) $(/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Current Socket' | tr -s ' ' | tr '\n' '|')"; sleep 0.05; done ) > $O/smi_trace.txt 2>&1 &
SMI=$!
sleep 1
{ echo "random-start $(date +%s.%N)"; CONV_MICRO_BURST=2000 timeout 100 $B/conv_bf16_micro conv3_2; echo "random-end zero-start $(date +%s.%N)"; CONV_MICRO_ZERO=1 CONV_MICRO_BURST=2000 timeout 100 $B/conv_bf16_micro conv3_2; echo "zero-end $(date +%s.%N)"; } > $O/conv_long.txt 2>&1
wait $SMI
cat $O/conv_long.txt; grep -c sclk $O/smi_trace.txt

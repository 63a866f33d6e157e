#!/bin/bash
# Round 5: what the matrix pipes sustain chip-wide on bf16 32x32x16 against 16x16x32, on random operands and with half of one operand zero (post-ReLU-like).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05o; mkdir -p $O
timeout 200 ./scripts/micro/_bin/mfma_peak_micro 1 20000 10 > $O/r05_mfma_peak_micro.txt 2>&1; cat $O/r05_mfma_peak_micro.txt

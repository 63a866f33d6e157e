#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r04c; mkdir -p $O; B=scripts/micro/_bin
{ for pr in 0 1 2; do echo "== form 2, FRCNN_BF16_PAIR_PRIO=$pr (0 none, 1 consumers first, 2 producers first)"; FRCNN_BF16_PAIR_PRIO=$pr timeout 60 $B/conv_pair_micro; done
  for rw in 6 4; do echo "== form 1 (one wave per SIMD, weights in registers), RW $rw"; FRCNN_BF16_PAIR_FORM=1 FRCNN_BF16_PAIR_RW=$rw timeout 60 $B/conv_pair_micro; done; } > $O/r04_conv_pair_micro.txt 2>&1; cat $O/r04_conv_pair_micro.txt

#!/bin/bash
# Round 3: weight-gradient kernel study on the torch-free harness (scripts/micro/wgrad_micro.cpp): single- / double-buffered form,
# timing ablations (separate -DFRCNN_TIMING_ABLATIONS library), then the training step.  (The three-workgroups-per-CU and wave-priority
# variants this script also timed in round 3 were removed from the kernel; profiles/r03_wgrad_ablation_micro.txt keeps their numbers.)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r03wg; mkdir -p $O
B=scripts/micro/_bin
run() { echo "## $*"; env "$@" 2>&1; }
{
echo "== default (per-layer pick of the single- / double-buffered form)"; $B/wgrad_micro
echo "== FRCNN_WGRAD_DB=0 (single buffer, 2 workgroups per CU, everywhere)"; FRCNN_WGRAD_DB=0 $B/wgrad_micro
echo "== FRCNN_WGRAD_DB=1";                           FRCNN_WGRAD_DB=1 $B/wgrad_micro conv1_2 conv3_2 conv4_2 conv5_1
for a in 1 4 8 5; do
  echo "== ablation build ABL=$a (1 no DMA, 4 no MFMA, 8 no slab stores, 5 = neither DMA nor MFMA)"; FRCNN_WGRAD_DB=0 FRCNN_WGRAD_ABL=$a $B/wgrad_micro_abl conv1_2 conv3_2 conv4_2 conv5_1
done
echo "== split-product kernel"; $B/wgrad_micro --f32s conv1_2 conv3_2 conv4_2 conv5_1
} > $O/wgrad_micro.txt 2>&1
cat $O/wgrad_micro.txt
timeout 300 python bench.py --mode train --steps 30 --warmup 3 > $O/train.json 2>/dev/null; echo "train rc=$?"; cut -c1-160 $O/train.json

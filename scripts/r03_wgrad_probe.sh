#!/bin/bash
# Round 3: weight-gradient kernel study on the torch-free harness (scripts/micro/wgrad_micro.cpp): workgroups per CU, wave priorities,
# double buffering, timing ablations (separate -DFRCNN_TIMING_ABLATIONS library), then the GPU tests of the training step with the candidate default.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r03wg; mkdir -p $O
B=scripts/micro/_bin
run() { echo "## $*"; env "$@" 2>&1; }
{
echo "== default (per-layer pick of the single- / double-buffered form)"; $B/wgrad_micro
echo "== FRCNN_WGRAD_DB=0 (single buffer, 2 workgroups per CU, everywhere)"; FRCNN_WGRAD_DB=0 $B/wgrad_micro
echo "== FRCNN_WGRAD_WPS=3";                          FRCNN_WGRAD_DB=0 FRCNN_WGRAD_WPS=3 $B/wgrad_micro
echo "== FRCNN_WGRAD_PRIO=1 (2 per CU)";              FRCNN_WGRAD_DB=0 FRCNN_WGRAD_PRIO=1 $B/wgrad_micro conv1_2 conv3_2 conv4_2 conv5_1
echo "== FRCNN_WGRAD_PRIO=1 FRCNN_WGRAD_WPS=3";       FRCNN_WGRAD_DB=0 FRCNN_WGRAD_PRIO=1 FRCNN_WGRAD_WPS=3 $B/wgrad_micro conv1_2 conv3_2 conv4_2 conv5_1
echo "== FRCNN_WGRAD_DB=1";                           FRCNN_WGRAD_DB=1 $B/wgrad_micro conv1_2 conv3_2 conv4_2 conv5_1
for w in 2 3; do for a in 1 4 8 5; do
  echo "== ablation build WPS=$w ABL=$a (1 no DMA, 4 no MFMA, 8 no slab stores, 5 = neither DMA nor MFMA)"; FRCNN_WGRAD_DB=0 FRCNN_WGRAD_WPS=$w FRCNN_WGRAD_ABL=$a $B/wgrad_micro_abl conv1_2 conv3_2 conv4_2 conv5_1
done; done
echo "== split-product kernel"; $B/wgrad_micro --f32s conv1_2 conv3_2 conv4_2 conv5_1
} > $O/wgrad_micro.txt 2>&1
cat $O/wgrad_micro.txt
echo "== GPU tests of the weight gradient with 3 workgroups per CU"
FRCNN_WGRAD_WPS=3 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wgrad or train" --timeout 600 > $O/pytest_wps3.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_wps3.log
for w in 2 3; do FRCNN_WGRAD_WPS=$w timeout 300 python bench.py --mode train --steps 30 --warmup 3 > $O/train_wps$w.json 2>/dev/null; echo "train WPS=$w rc=$?"; cut -c1-160 $O/train_wps$w.json; done

#!/bin/bash
# SQ counters of the weight-gradient kernel (conv3_2 shape) in three forms: full, without DMA, without MFMAs (ablation library)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r03wg; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*" | sort -u | tr '\n' ' ' ) > $O/avail_counters.txt 2>&1
for a in 0 1 4; do
  echo "=== FRCNN_WGRAD_ABL=$a"
  ROI_BIN="env FRCNN_WGRAD_ABL=$a WGRAD_MICRO_BURST=2 $R/scripts/micro/_bin/wgrad_micro_abl" scripts/micro/roi_pmc.sh conv_wgrad_dma conv3_2
done > $O/wgrad_pmc.txt 2>&1
cat $O/wgrad_pmc.txt; wc -c $O/avail_counters.txt

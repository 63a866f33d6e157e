#!/bin/bash
# Round 5, GPU call: the arg-max form with the paired scan against the one-RoI-at-a-time scan.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05m; mkdir -p $O
ROI_MICRO_BURST=50 timeout 300 ./scripts/micro/_bin/roi_micro DEFAULT=1 > $O/r05_roi_micro.txt 2>&1; cat $O/r05_roi_micro.txt
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -k "roi" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_subset.log

#!/bin/bash
# Round 5, GPU call 1: the RoI-pooling backward's new forms (NCH channels per workgroup x DEPTH RoIs in flight per wave) against the round-3 four-channel kernel,
# torch-free (scripts/micro/roi_micro: graphs of ten launches), then the GPU tests the tuning-registry refactor and the DPP fix touch.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05a; mkdir -p $O
ROI_MICRO_BURST=50 timeout 300 ./scripts/micro/_bin/roi_micro DEFAULT=1 > $O/r05_roi_micro.txt 2>&1; cat $O/r05_roi_micro.txt
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "roi or nms or conv1_pair or wgrad_forms or default_picks or library or proposals" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_subset.log

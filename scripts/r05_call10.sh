#!/bin/bash
# Round 5, GPU call: the full-size stage-2 training step against the float64 arbiter (new test), and the NMS / proposal tests after the revert.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -s --timeout 1200 -k "rcnn_train_step_600x1000 or nms_staged or proposals_golden" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; grep -E "^PARITY|passed|failed|Error|error" $O/pytest_subset.log | cut -c1-1500

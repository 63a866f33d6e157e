#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/r02t
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "train_step" --timeout 1000 > gpurun_out/r02t/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed|PARITY rpn|full-size weight" gpurun_out/r02t/pytest.log | cut -c1-1200

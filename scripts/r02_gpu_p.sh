#!/bin/bash
# Round 2: RPN training step with forward / input-gradient convolutions as split products: parity + timing A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02p}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest train split"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "train_step" --timeout 800 > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -i "error\|assert" $O/pytest.log | head -5



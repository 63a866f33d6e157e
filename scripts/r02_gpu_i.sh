#!/bin/bash
# Round 2: RoI cells kernel with per-workgroup geometry tables + LPT order -- parity and timing.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02i}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest roi"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "roi or forward_600" --timeout 600 > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
echo "== roi ablate"; timeout 300 python scripts/roi_ablate.py 2>&1 | grep -v amdgpu.ids | tail -16

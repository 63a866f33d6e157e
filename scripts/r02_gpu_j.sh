#!/bin/bash
# Round 2: split-bf16 fp32 convolution -- parity + per-layer timing against the native fp32 MFMA kernel.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02j}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest f32s"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "f32s" --timeout 500 > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
echo "== bench"; XCD_AB=1 timeout 600 python scripts/conv_f32s_bench.py > $O/f32s_bench.log 2>&1; grep -v amdgpu.ids $O/f32s_bench.log | tail -30

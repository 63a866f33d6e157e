#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02r}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wgrad_f32s or train_step" --timeout 800 2>&1 | tail -3
echo "== wgrad bench"; SPLITS=0,512 timeout 600 python scripts/wgrad_bench.py 2>&1 | grep -v amdgpu.ids | tail -12
echo "== bench train split"; timeout 600 python bench.py --mode train --dtype f32s --steps 20 --warmup 3 > $O/bench_train_f32s.json 2>> $O/bench.err; cut -c1-250 $O/bench_train_f32s.json

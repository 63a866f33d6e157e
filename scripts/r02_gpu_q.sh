#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02q}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "train_step or conv1_ or weight_packs" --timeout 800 2>&1 | tail -2
echo "== bench train split"; timeout 600 python bench.py --mode train --dtype f32s --steps 20 --warmup 3 > $O/bench_train_f32s.json 2>> $O/bench.err; cut -c1-250 $O/bench_train_f32s.json
echo "== bench train mfma"; timeout 600 python bench.py --mode train --steps 20 --warmup 3 > $O/bench_train.json 2> $O/bench.err; cut -c1-250 $O/bench_train.json

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out/r02w; O=gpurun_out/r02w
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "maxpool or conv_backward or train_step or bias" --timeout 800 2>&1 | tail -2
timeout 600 python bench.py --mode train --steps 20 --warmup 3 > $O/bench_train.json 2> $O/bench.err; cut -c1-200 $O/bench_train.json
timeout 600 python bench.py --mode train --dtype f32s --steps 20 --warmup 3 > $O/bench_train_f32s.json 2>> $O/bench.err; cut -c1-200 $O/bench_train_f32s.json

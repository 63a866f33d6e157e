#!/bin/bash
# Round 2: split-bf16 fp32 convolutions end to end -- full-size parity + bench line.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02k}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest f32s"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -s -k "f32s" --timeout 800 > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep "^PARITY" $O/pytest.log | cut -c1-1500
echo "== bench f32s"; timeout 600 python bench.py --dtype f32s --steps 50 --warmup 5 > $O/bench_f32s.json 2> $O/bench.err; echo "rc=$?"; cut -c1-2600 $O/bench_f32s.json; tail -3 $O/bench.err

#!/bin/bash
# Round 2, validation run after the hardware bf16 pack conversion: GPU suite + bf16 line + RoI A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02g}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2
echo "== bench bf16"; timeout 600 python bench.py --dtype bf16 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench.err; echo "rc=$?"; cut -c1-200 $O/bench_bf16.json
echo "== prop bench"; timeout 300 python scripts/prop_bench.py > $O/prop.log 2>&1; tail -15 $O/prop.log

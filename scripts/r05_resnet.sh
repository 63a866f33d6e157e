#!/bin/bash
# Round 5: BASELINE configs[3] (ResNet-101 trunk, 1000 / 300 proposals) re-measured on the final tree.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05v; mkdir -p $O
timeout 600 python scripts/resnet_bench.py > $O/r05_bench_resnet101.json 2> $O/err.log; echo "rc=$?"; cut -c1-400 $O/r05_bench_resnet101.json | tail -1

#!/bin/bash
# Round 5: BASELINE configs[3] (ResNet-101 trunk, 1000 / 300 proposals) re-measured on the final tree.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05w; mkdir -p $O
timeout 600 python scripts/resnet_bench.py > $O/r05_bench_resnet101.json 2> $O/err.log; echo "rc=$?"; python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith(chr(123))][-1]); print(d[\"value\"], d[\"two_images_in_flight\"])" $O/r05_bench_resnet101.json

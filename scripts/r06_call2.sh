#!/bin/bash
# Round 6, call 2: the RoI / edge tests on the hardware and the three non-contract bench lines with their new `roofline` / `cpu_baseline` objects.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/${TAG:-r06b}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "roi or edge or nms" --timeout 600 > $O/pytest_roi.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_roi.log
timeout 900 python bench.py --mode train --steps 40 --warmup 3 > $O/r06_bench_train.json 2> $O/bench.err; echo "train rc=$?"; cut -c1-200 $O/r06_bench_train.json
timeout 900 python bench.py --mode train-rcnn --steps 20 --warmup 3 > $O/r06_bench_train_rcnn_device.json 2>> $O/bench.err; echo "train-rcnn rc=$?"; cut -c1-200 $O/r06_bench_train_rcnn_device.json
timeout 900 python scripts/resnet_bench.py > $O/r06_bench_resnet101.json 2>> $O/bench.err; echo "resnet rc=$?"; cut -c1-200 $O/r06_bench_resnet101.json
tail -5 $O/bench.err

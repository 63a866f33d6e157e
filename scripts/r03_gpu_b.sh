#!/bin/bash
# round 3: full-size tests added this round + the config-4 (ResNet-101) measurement record with its rocprof kernel statistics
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "image_to_detections or rpn_train_step_600x1000" > gpurun_out/r03_fullsize_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_fullsize_tests.txt
grep "^PARITY" gpurun_out/r03_fullsize_tests.txt | cut -c1-1500
timeout 600 python scripts/resnet_bench.py > gpurun_out/r03_bench_resnet101.json 2> gpurun_out/r03_bench_resnet101.err; echo "resnet rc=$?"; tail -2 gpurun_out/r03_bench_resnet101.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rprof_rn
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rprof_rn -o rn -- python "$R/scripts/resnet_bench.py" > /tmp/rn.log 2>&1; echo "rocprof rc=$?"
cd "$R"
f=$(find /tmp/rprof_rn -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03_resnet101_kernel_stats.csv && head -12 gpurun_out/r03_resnet101_kernel_stats.csv | cut -c1-200
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03_bench_resnet101.json") if l.startswith('{"metric"')][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["trunk_ms"], d["fc6"], {k: v for k, v in d["stages_ms"].items() if v > 0.05})
PY

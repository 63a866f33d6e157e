#!/bin/bash
# Round 2: full GPU suite + the default bench line (native fp32 contract line + split-product variant + parity + CPU baseline).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02m}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2
grep -E "^PARITY|full-size weight" $O/pytest_gpu.log > $O/parity_reports.txt
echo "== bench default"; timeout 900 python bench.py > $O/bench_default.json 2> $O/bench.err; echo "rc=$?"; cut -c1-300 $O/bench_default.json; tail -2 $O/bench.err
python - <<'PY'
import json,sys
d=json.load(open("gpurun_out/%s/bench_default.json" % "r02m"))
print(json.dumps(d.get("f32_split_products"))[:1500])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["nms_roi"]["proposals_nms_us"], d["nms_roi"]["roi_pool_us"], d["parity"]["ok"])
PY

#!/bin/bash
# Round 4, call 2: the repaired K-split test, the RoI / stage-2 tests after the NaN-flag and linear-dx changes, and the first stage-2 step measurement
# (dx = dy W through the 1x1 convolution reading W as stored vs through a transposed copy of W).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s --timeout 600 -k "staging or roi or rcnn_train or default_picks or strip" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_sel.log | tail -2
grep -E "^PARITY" $O/pytest_sel.log | head -20
for dx in conv transpose; do
  FRCNN_LINEAR_DX=$dx timeout 600 python bench.py --mode train-rcnn --steps 20 --warmup 3 > $O/r04_bench_train_rcnn_$dx.json 2> $O/train_rcnn_$dx.err; echo "train-rcnn $dx rc=$?"
  python - "$O/r04_bench_train_rcnn_$dx.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
    print(round(d["ms_per_step"], 3), "ms/step", d["stages_ms"], d["losses"])
except Exception as e:
    print("no line:", e)
PY
done
tail -5 $O/train_rcnn_conv.err

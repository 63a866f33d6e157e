#!/bin/bash
# Round 4: the stage-2 step measured (dx = dy W through the 1x1 convolution reading W as stored vs through a transposed copy of W).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r04b; mkdir -p $O
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

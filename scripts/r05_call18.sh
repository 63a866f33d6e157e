#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05s8; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "rcnn or dropout" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log
timeout 300 python bench.py --mode train-rcnn --dropout-rng device --steps 20 --warmup 3 > $O/train_rcnn.json 2>> $O/err.log
python - <<PY
import json
d=json.loads([l for l in open("$O/train_rcnn.json") if l.startswith('{"metric"')][-1])
print("train-rcnn", round(d["ms_per_step"],3)); print(" gpu ", d["stages_ms"]); print(" host", d["host_enqueue_ms"])
PY
timeout 300 python bench.py --mode train --steps 30 --warmup 3 > $O/train.json 2>> $O/err.log
python - <<PY
import json
d=json.loads([l for l in open("$O/train.json") if l.startswith('{"metric"')][-1])
print("train", round(d["ms_per_step"],3), d.get("stages_ms"))
PY

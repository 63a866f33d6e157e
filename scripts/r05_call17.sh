#!/bin/bash
# Stage-2 step: head backward on the kept rows + ProposalTargetLayer's sampling under the head's forward pass.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05s2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "rcnn or dropout or target" -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|RCNN_BWD" $O/pytest.log | tail -8
for rows in kept all; do
  FRCNN_RCNN_BWD_ROWS=$rows timeout 300 python bench.py --mode train-rcnn --dropout-rng device --steps 20 --warmup 3 > $O/train_rcnn_$rows.json 2>> $O/err.log
  python - <<PY
import json
d=json.loads([l for l in open("$O/train_rcnn_$rows.json") if l.startswith('{"metric"')][-1])
print("$rows", round(d["ms_per_step"],3), d["stages_ms"], d["config"].get("head_backward_rows_last_step"))
PY
done

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r04t; mkdir -p $O
python -c "import torch; print('priority range (least, greatest):', torch.cuda.Stream.priority_range())"
for pr in default 2 1 0 -1; do
  if [ $pr = default ]; then unset FRCNN_SIDE_STREAM_PRIO; else export FRCNN_SIDE_STREAM_PRIO=$pr; fi
  timeout 300 python bench.py --mode train --steps 40 --warmup 3 > $O/train_prio_$pr.json 2> $O/train_prio_$pr.err; echo -n "side-stream priority $pr rc=$? : "
  python - "$O/train_prio_$pr.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]); print(round(d["ms_per_step"], 4), "ms/step; without the ProposalLayer", round(d["ms_per_step_without_proposal_layer"], 4))
except Exception as e: print("no line:", e)
PY
done

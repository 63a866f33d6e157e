#!/bin/bash
# Per-kernel statistics of the stage-2 training step.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=$R/gpurun_out/r05s6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rcnn -o rcnn -- python $R/bench.py --mode train-rcnn --dropout-rng device --steps 20 --warmup 3 > $O/bench.log 2>&1
f=$(find /tmp/prof_rcnn -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_rcnn_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/train_rcnn_kernel_stats.csv")))
for r in rows[:40]:
    print("%-110s %5s %9.1f %6.2f"%(r['Name'][:110],r['Calls'],float(r['AverageNs'])/1e3,float(r['Percentage'])))
PY

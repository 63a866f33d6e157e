#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r04o; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -s --timeout 600 -k "roi or rcnn_train" > $O/pytest_roi.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_roi.log | tail -2
for sp in default 0; do
  if [ $sp = default ]; then unset FRCNN_ROI_BWD_SPLIT; else export FRCNN_ROI_BWD_SPLIT=$sp; fi
  timeout 300 python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-split-variant --no-bf16-variant > $O/bench_split_$sp.json 2> $O/bench_$sp.err; echo -n "FRCNN_ROI_BWD_SPLIT=$sp: "
  python - "$O/bench_split_$sp.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]); s = d["roofline"]["secondary"]
print({k: s[k] for k in ("roi_pool_us", "roi_pool_fwd_argmax_us", "roi_pool_bwd_us", "roi_pool_bwd_frac_of_hbm_peak")})
PY
done
unset FRCNN_ROI_BWD_SPLIT
timeout 300 python bench.py --mode train-rcnn --steps 20 --warmup 3 > $O/train_rcnn.json 2>> $O/bench.err; python - "$O/train_rcnn.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]); print(round(d["ms_per_step"], 3), d["stages_ms"])
PY

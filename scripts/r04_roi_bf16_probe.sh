#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r04q; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -s --timeout 600 -k "roi or vgg16_forward_600x1000_bf16 or image_to_detections or bf16_inference" > $O/pytest_roi.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_roi.log | tail -2
for k in quads cells; do
  if [ $k = quads ]; then unset FRCNN_ROI_KERNEL; else export FRCNN_ROI_KERNEL=cells; fi
  timeout 300 python bench.py --dtype bf16 --steps 100 --warmup 5 --no-cpu-baseline > $O/bench_bf16_$k.json 2> $O/bench_$k.err; echo -n "bf16 line, RoI stage on the $k kernel: "
  python - "$O/bench_bf16_$k.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1]); print(round(d["value"], 1), "img/s", round(d["ms_per_step"], 4), "ms; roi stage (eager event)", d["stages_ms"].get("roi_pool"))
PY
done

#!/bin/bash
# Round 5, GPU call 8: the fused NMS launch (mask + scan + second stage in one) against the three-launch chain; NMS / proposal GPU tests; the bench's feed variant.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05h; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -k "nms or proposal or detections or captured" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_subset.log
timeout 300 python scripts/prop_bench.py > $O/r05_prop_bench.txt 2>&1; cat $O/r05_prop_bench.txt
timeout 600 python bench.py --no-split-variant > $O/r05_bench_feed.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/r05_bench_feed.json; tail -5 $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05h/r05_bench_feed.json") if l.startswith('{"metric"')][-1])
print("value", d["value"], "with_feed", d.get("with_feed"))
print("bf16", (d.get("bf16_config3") or {}).get("value"), (d.get("bf16_config3") or {}).get("with_feed"))
print("secondary", d["roofline"].get("secondary"))
PY

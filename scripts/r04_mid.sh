#!/bin/bash
# Round 4, mid-round check: the whole GPU suite on the current tree + the bf16 line (conv1 pair launch default) + rocprof of it.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r04m; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -q -s --timeout 1200 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head
grep -E "^PARITY|full-size weight" $O/pytest_gpu.log > $O/parity_reports.txt
timeout 600 python bench.py --dtype bf16 --steps 100 --warmup 5 --no-cpu-baseline > $O/r04_bench_bf16.json 2> $O/bench.err; echo "bench bf16 rc=$?"
python - "$O/r04_bench_bf16.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
print(round(d["value"], 1), "img/s", round(d["ms_per_step"], 4), "ms; conv", round(d["roofline"]["conv_ms_per_image"], 4), "ms frac", round(d["roofline"]["frac"], 4))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_bf16" -o r04_bf16 -- python "$R/bench.py" --dtype bf16 --steps 100 --warmup 5 --no-cpu-baseline > "$R/$O/prof_bf16.log" 2>&1; echo "rocprof bf16 rc=$?"

#!/bin/bash
# kernel-level profile of the bf16 line (graph replay): where the step really goes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bf16" -o r01_bf16 -- python "$R/bench.py" --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > "$R/gpurun_out/prof_bf16.log" 2>&1; echo "rc=$?"
cd "$R"; tail -n 1 gpurun_out/prof_bf16.log | cut -c1-200
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_bf16/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-90s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY

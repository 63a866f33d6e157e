#!/bin/bash
# Targeted evidence refresh after a change that only touches the bf16 line: its parity tests, bench line and rocprofv3 kernel stats.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02z5}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "bf16 or roi_pool or smoke or abi" > $O/pytest_bf16.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_bf16.log
timeout 300 python bench.py --dtype bf16 --steps 50 --warmup 5 > $O/r02_bench_bf16.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/r02_bench_bf16.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_bf16" -o r02_bf16 -- python "$R/bench.py" --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > "$R/$O/prof_bf16.log" 2>&1; echo "rocprof bf16 rc=$?"

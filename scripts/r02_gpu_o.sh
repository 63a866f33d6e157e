#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r02o}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1; do
for x in 0; do
  FRCNN_F32S_XCD=$x timeout 600 python bench.py --dtype f32s --steps 100 --warmup 5 --no-cpu-baseline > $O/bench_f32s_xcd$x.json 2>> $O/bench.err
  python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench_f32s_xcd$x.json")); print("xcd=$x", round(d["value"],1), round(d["ms_per_step"],4), round(d["roofline"]["conv_ms_per_image"],4), d["stages_ms"])
PY
done; done

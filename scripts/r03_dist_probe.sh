#!/bin/bash
# Round 3, VERDICT item 3: the N > 1 code path on the one GPU of a development box.
#   1. RCCL at world size 1 (bench.py --mode train --dist-world1): init + three bucketed async all-reduces + stream waits really execute
#   2. two ranks sharing the GPU over gloo (staged through pinned host memory), with host-side comm timing (FRCNN_COMM_TRACE=1)
#   3. the same two ranks in inference mode
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python bench.py --mode train --dist-world1 --steps 40 > gpurun_out/r03_bench_nccl_w1_train.json 2> gpurun_out/r03_bench_nccl_w1_train.err; echo "nccl w1 rc=$?"; tail -2 gpurun_out/r03_bench_nccl_w1_train.err
timeout 400 python bench.py --mode train --steps 40 > gpurun_out/r03_bench_train.json 2> gpurun_out/r03_bench_train.err; echo "single rc=$?"
FRCNN_COMM_TRACE=1 timeout 600 python bench.py --gpus 2 --mode train --steps 20 > gpurun_out/r03_bench_2rank_gloo_train.json 2> gpurun_out/r03_bench_2rank_gloo_train.err; echo "gloo2 train rc=$?"; tail -3 gpurun_out/r03_bench_2rank_gloo_train.err
timeout 600 python bench.py --gpus 2 --steps 100 --no-cpu-baseline > gpurun_out/r03_bench_2rank_gloo_infer.json 2> gpurun_out/r03_bench_2rank_gloo_infer.err; echo "gloo2 infer rc=$?"
python - <<'PY'
import json
for f in ("r03_bench_nccl_w1_train", "r03_bench_train", "r03_bench_2rank_gloo_train", "r03_bench_2rank_gloo_infer"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.1f  ms/step %.3f" % (d["value"], d["ms_per_step"]), d.get("stages_ms"), d.get("dist"), d.get("per_rank"))
    except Exception as e:
        print(f, "unreadable:", e)
PY

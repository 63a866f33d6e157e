#!/bin/bash
# The round's evidence run (P = file-name prefix, default r05): GPU suite (PARITY lines kept), the contract line (with nms_roi incl. the training forms, bf16_config3,
# f32_split_products, parity, CPU baseline), the bf16 / f32s / training lines, RCCL at world size 1 and the 2-rank gloo smoke,
# rocprofv3 kernel statistics of the bench commands, PMC passes (HBM traffic per dtype; SQ counters of the RoI kernel through the
# torch-free harness; MFMA counters of conv3_2 in bf16), the store / launch micro-measurements behind DESIGN 3.2.
# STAGES selects a subset: "tests bench prof pmc dist micro" (default: all).
set -u
P=${P:-r05}                     # file-name prefix of the round
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${TAG:-r05z}; STAGES=${STAGES:-"tests bench prof pmc dist micro"}
cd "$R"; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
has() { case " $STAGES " in *" $1 "*) return 0;; *) return 1;; esac; }
if has tests; then
  echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
  echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q -s --timeout 1200 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2
  grep -E "^PARITY|full-size weight" $O/pytest_gpu.log > $O/parity_reports.txt
fi
if has bench; then
  echo "== bench f32 (contract line)"; T0=$(date +%s); timeout 900 python bench.py > $O/${P}_bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"; cut -c1-300 $O/${P}_bench.json
  echo "== bench bf16"; timeout 600 python bench.py --dtype bf16 --steps 100 --warmup 5 > $O/${P}_bench_bf16.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-200 $O/${P}_bench_bf16.json
  echo "== bench f32s"; timeout 600 python bench.py --dtype f32s --steps 100 --warmup 5 > $O/${P}_bench_f32s.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-200 $O/${P}_bench_f32s.json
  echo "== bench train"; timeout 600 python bench.py --mode train --steps 40 --warmup 3 > $O/${P}_bench_train.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-200 $O/${P}_bench_train.json
  echo "== bench train f32s"; timeout 600 python bench.py --mode train --dtype f32s --steps 40 --warmup 3 > $O/${P}_bench_train_f32s.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-200 $O/${P}_bench_train_f32s.json
fi
if has dist; then
  echo "== RCCL world 1 / gloo 2 ranks"
  timeout 400 python bench.py --mode train --dist-world1 --steps 40 > $O/${P}_bench_nccl_w1_train.json 2> $O/dist.err; echo "nccl w1 rc=$?"
  FRCNN_COMM_TRACE=1 timeout 600 python bench.py --gpus 2 --mode train --steps 20 > $O/${P}_bench_2rank_gloo_train.json 2>> $O/dist.err; echo "gloo2 train rc=$?"
  timeout 600 python bench.py --gpus 2 --steps 100 --no-cpu-baseline > $O/${P}_bench_2rank_gloo_infer.json 2>> $O/dist.err; echo "gloo2 infer rc=$?"
fi
if has prof; then
  echo "== rocprof kernel stats"; cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o ${P} -- python "$R/bench.py" --steps 100 --warmup 5 --no-cpu-baseline --no-split-variant --no-bf16-variant --no-feed-variant --no-two-streams-variant > "$R/$O/prof.log" 2>&1; echo "rocprof f32 rc=$?"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_bf16" -o ${P}_bf16 -- python "$R/bench.py" --dtype bf16 --steps 100 --warmup 5 --no-cpu-baseline --no-feed-variant --no-two-streams-variant > "$R/$O/prof_bf16.log" 2>&1; echo "rocprof bf16 rc=$?"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_train" -o ${P}_train -- python "$R/bench.py" --mode train --steps 10 --warmup 2 > "$R/$O/prof_train.log" 2>&1; echo "rocprof train rc=$?"
  cd "$R"
  for d in prof prof_bf16 prof_train; do find $O/$d -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null; done       # keep the per-kernel statistics, drop the raw traces
fi
if has pmc; then
  echo "== hbm traffic PMC"; cd /tmp && export TMPDIR=/tmp
  for dt in f32 bf16; do for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/$O/traffic_${dt}_$c" -o t -- python "$R/bench.py" --dtype $dt --steps 3 --warmup 2 --no-cpu-baseline --no-split-variant --no-bf16-variant --no-feed-variant --no-two-streams-variant > "$R/$O/traffic_${dt}_$c.log" 2>&1; echo "$dt $c rc=$?"
  done; done
  cd "$R"; for dt in f32 bf16; do python scripts/gpu_traffic.py $O $dt ${P} > $O/traffic_${dt}_summary.txt 2>&1; done; tail -8 $O/traffic_f32_summary.txt
  rm -rf $O/traffic_f32_FETCH_SIZE $O/traffic_f32_WRITE_SIZE $O/traffic_bf16_FETCH_SIZE $O/traffic_bf16_WRITE_SIZE        # raw counter csvs: tens of MB (gpurun copies back at most 64 MiB); the summaries stay
  echo "== RoI SQ counters (torch-free harness)"
  scripts/micro/roi_pmc.sh quads DEFAULT=1 > $O/${P}_roi_pmc.txt 2>&1; tail -20 $O/${P}_roi_pmc.txt
  echo "== bf16 conv3_2 MFMA counters"; cd /tmp
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/mfma_bf16" -o m -- bash -c "cd $R && ./scripts/micro/_bin/conv_bf16_micro conv3_2" > "$R/$O/mfma_bf16.log" 2>&1; echo "mfma pmc rc=$?"
  cd "$R"
  python - "$O" "$P" <<'PY'
import csv, glob, sys, collections, json
O, P = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(O + "/mfma_bf16/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_dma_bf16_kernel" in r["Kernel_Name"] or "conv_strip_bf16_kernel" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
out = {k: {"per_launch_mean": v / max(len(ids), 1), "launches": len(ids)} for k, (v, ids) in acc.items()}
out["_note"] = "the default pick for the conv3_2 shape (256 -> 256, 150 x 250: conv_strip_bf16_kernel, form D with direct stores) through scripts/micro/conv_bf16_micro; rocprofv3 --pmc, one pass.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)"
json.dump(out, open(O + "/" + P + "_mfma_pmc_summary.json", "w"), indent=1, sort_keys=True)
print({k: v["per_launch_mean"] for k, v in out.items() if k != "_note"})
PY
  rm -rf $O/mfma_bf16
fi
if has micro; then
  echo "== store / launch micro-measurements"
  ./scripts/micro/_bin/store_micro > $O/${P}_store_micro.txt 2>&1; head -8 $O/${P}_store_micro.txt
  ROI_MICRO_BURST=50 ./scripts/micro/_bin/roi_micro DEFAULT=1 FRCNN_ROI_ST=0 FRCNN_ROI_KERNEL=cells > $O/${P}_roi_micro.txt 2>&1; cat $O/${P}_roi_micro.txt
  ./scripts/micro/_bin/conv_bf16_micro > $O/${P}_conv_bf16_micro.txt 2>&1; tail -12 $O/${P}_conv_bf16_micro.txt
  ./scripts/micro/_bin/conv_f32_micro > $O/${P}_conv_f32_micro.txt 2>&1; tail -3 $O/${P}_conv_f32_micro.txt
  { ./scripts/micro/_bin/wgrad_micro; echo "== split products"; ./scripts/micro/_bin/wgrad_micro --f32s; echo "== conv1_1, generic kernel"; FRCNN_WGRAD_CONV1=generic ./scripts/micro/_bin/wgrad_micro conv1_1; } > $O/${P}_wgrad_micro.txt 2>&1; tail -14 $O/${P}_wgrad_micro.txt
  timeout 200 ./scripts/micro/_bin/mfma_dma_micro > $O/${P}_mfma_filler_micro.txt 2>&1; head -8 $O/${P}_mfma_filler_micro.txt
  ./scripts/micro/_bin/dma_align_micro > $O/${P}_dma_align_micro.txt 2>&1; head -4 $O/${P}_dma_align_micro.txt
fi
echo "== done"

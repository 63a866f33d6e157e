#!/bin/bash
# Round 6, the evidence run on the final tree: everything scripts/gpu_evidence.sh collects under the r06 prefix (smoke, the whole GPU suite without -x, the contract
# line -- now with the f16_config3 block --, the bf16 / f32s / training lines, RCCL at world size 1 / gloo with two ranks, rocprofv3 kernel statistics of the f32 and
# bf16 commands (one-at-a-time processes: --no-two-streams-variant --no-feed-variant), HBM-traffic and RoI / bf16 MFMA PMC passes, the micro harnesses) plus what
# VERDICT r05 asked for: `roofline` + `cpu_baseline` on the training / stage-2 / ResNet-101 lines, an MFMA-busy counter pass over the SHIPPED fp32 convolution
# kernel (with the busy fraction written into the file), the fp16 line with its parity block, the FC weight-stream kernel against round 5's, package power beside
# the bench lines.  Outputs under gpurun_out/$TAG/; scripts/collect_profiles.py <TAG> r06 copies what is judged into profiles/r06_*.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
START=$(date +%s); STAGES="${STAGES:-tests bench prof pmc dist micro}" TAG=${TAG:=r06z} P=r06 bash scripts/gpu_evidence.sh
O=gpurun_out/$TAG; B=scripts/micro/_bin
has() { case " ${EXTRA:-lines f16 fc mfma resnet power} " in *" $1 "*) return 0;; *) return 1;; esac; }
if has lines; then
  timeout 900 python bench.py --mode train-rcnn --dropout-rng device --steps 20 --warmup 3 > $O/r06_bench_train_rcnn_device.json 2>> $O/bench.err; echo "train-rcnn rc=$?"; cut -c1-160 $O/r06_bench_train_rcnn_device.json | tail -1
  timeout 600 python bench.py --mode train-rcnn --dtype f32s --dropout-rng device --steps 20 --warmup 3 --no-cpu-baseline > $O/r06_bench_train_rcnn_f32s.json 2>> $O/bench.err; echo "train-rcnn f32s rc=$?"
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_rcnn" -o r06_train_rcnn -- python "$R/bench.py" --mode train-rcnn --dropout-rng device --steps 10 --warmup 2 --no-cpu-baseline > "$R/$O/prof_rcnn.log" 2>&1; echo "rocprof train-rcnn rc=$?" ); find $O/prof_rcnn -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
fi
if has f16; then
  timeout 600 python bench.py --dtype f16 --steps 100 --warmup 5 > $O/r06_bench_f16.json 2>> $O/bench.err; echo "bench f16 rc=$?"; cut -c1-200 $O/r06_bench_f16.json | tail -1
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_f16" -o r06_f16 -- python "$R/bench.py" --dtype f16 --steps 100 --warmup 5 --no-cpu-baseline --no-feed-variant --no-two-streams-variant > "$R/$O/prof_f16.log" 2>&1; echo "rocprof f16 rc=$?" ); find $O/prof_f16 -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
fi
if has fc; then
  { timeout 200 $B/linear_bf16_micro; } > $O/r06_linear_bf16_micro_final.txt 2>&1; grep -E "^\[(old|ring)" $O/r06_linear_bf16_micro_final.txt | head -8
fi
if has mfma; then
  # MFMA-busy counters of the SHIPPED fp32 convolution kernel (north_star: "MFMA utilisation on the conv stack against chip peak"): conv3_2 and conv4_2 through the
  # torch-free harness, one rocprofv3 --pmc pass; busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), written into the file
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/mfma_f32" -o m -- bash -c "cd $R && ./scripts/micro/_bin/conv_f32_micro conv3_2 conv4_2 conv5_1" > "$R/$O/mfma_f32.log" 2>&1; echo "mfma f32 pmc rc=$?" )
  python - "$O" <<'PY'
import csv, glob, re, sys, collections, json
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
for f in glob.glob(O + "/mfma_f32/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_mfma_f32_kernel" in r["Kernel_Name"]:
            m = re.search(r"conv_mfma_f32_kernel<[^>]*>", r["Kernel_Name"])
            key = (m.group(0) if m else "conv_mfma_f32_kernel") + " grid " + r.get("Grid_Size", "?") + " lds " + r.get("LDS_Block_Size", "?")
            a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
out = {}
for key, ctrs in acc.items():
    d = {k: {"per_launch_mean": v / max(len(ids), 1), "launches": len(ids)} for k, (v, ids) in ctrs.items()}
    try:
        d["_mfma_busy_fraction"] = d["SQ_VALU_MFMA_BUSY_CYCLES"]["per_launch_mean"] / (d["GRBM_GUI_ACTIVE"]["per_launch_mean"] / 8.0 * 1024.0)
    except KeyError:
        pass
    out[key] = d
out["_note"] = ("conv_mfma_f32_kernel (the shipped fp32 convolution: LDS-DMA staging, stream-K) on the conv3_2 / conv4_2 / conv5_1 shapes through scripts/micro/conv_f32_micro, "
                "one rocprofv3 --pmc pass; keyed by template instantiation + grid size (one entry per layer shape).  _mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / "
                "(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)")
json.dump(out, open(O + "/r06_mfma_pmc_summary_f32.json", "w"), indent=1, sort_keys=True)
print({k: round(v.get("_mfma_busy_fraction", -1), 3) for k, v in out.items() if k != "_note"})
PY
  rm -rf $O/mfma_f32
  # ... and the same for the bf16 strip kernel, with the busy fraction written in (r05's file dropped the field)
  python - "$O" <<'PY'
import json, sys
O = sys.argv[1]
p = O + "/r06_mfma_pmc_summary.json"
try:
    d = json.load(open(p))
    d["_mfma_busy_fraction"] = d["SQ_VALU_MFMA_BUSY_CYCLES"]["per_launch_mean"] / (d["GRBM_GUI_ACTIVE"]["per_launch_mean"] / 8.0 * 1024.0)
    json.dump(d, open(p, "w"), indent=1, sort_keys=True)
    print("bf16 conv3_2 MFMA busy", round(d["_mfma_busy_fraction"], 3))
except Exception as e:
    print("bf16 mfma summary:", e)
PY
fi
if has resnet; then
  timeout 900 python scripts/resnet_bench.py > $O/r06_bench_resnet101.json 2>> $O/bench.err; echo "resnet rc=$?"; cut -c1-160 $O/r06_bench_resnet101.json | tail -1
fi
if has power; then
  O=$O/power bash scripts/bench_power.sh > $O/r06_bench_power.txt 2>&1; tail -12 $O/r06_bench_power.txt
fi
grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20
echo "whole evidence run: $(( $(date +%s) - START )) s"

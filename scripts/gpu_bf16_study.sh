#!/bin/bash
# PMC counters for the bf16 conv kernels (separate --pmc passes, kernel-trace only).  MODE = FRCNN_BF16_DMA value (-1 = default picks).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
export FRCNN_BF16_DMA=${MODE:--1}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d "$R/gpurun_out/bpmc1" -o p1 -- python "$R/scripts/conv_bf16_sweep.py" > "$R/gpurun_out/bpmc1.log" 2>&1; echo "rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES --output-format csv -d "$R/gpurun_out/bpmc2" -o p2 -- python "$R/scripts/conv_bf16_sweep.py" > "$R/gpurun_out/bpmc2.log" 2>&1; echo "rc=$?"
cd "$R"
python - <<'PY'
import csv, glob, collections
for d in ("bpmc1", "bpmc2"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "conv_" in r["Kernel_Name"] and "bf16" in r["Kernel_Name"]:
                acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for g in sorted(acc, key=int):
            print(d, "grid", g, " ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in acc[g].items()))
PY

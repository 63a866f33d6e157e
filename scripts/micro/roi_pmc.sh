#!/bin/bash
# SQ counters of the RoI kernels through the torch-free harness (two --pmc passes, --kernel-trace only).  Usage (GPU box):
#   scripts/micro/roi_pmc.sh <kernel-name-substring> [harness settings ...]      -> gpurun_out/roi_pmc_<tag>.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
PAT=$1; shift
BIN=${ROI_BIN:-$R/scripts/micro/_bin/roi_micro}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp1 /tmp/rp2
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d /tmp/rp1 -o p1 -- bash -c "cd $R && $BIN $*" > /tmp/rp1.log 2>&1; echo "rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d /tmp/rp2 -o p2 -- bash -c "cd $R && $BIN $*" > /tmp/rp2.log 2>&1; echo "rc=$?"
cd "$R"
python - "$PAT" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
for d in ("/tmp/rp1", "/tmp/rp2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, set()])
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
        for k, (v, ids) in sorted(acc.items()):
            print("%-24s per launch %14.1f   (%d launches)" % (k, v / max(len(ids), 1), len(ids)))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        ds = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
        if ds:
            ds.sort(); print("kernel duration ns: median %d  min %d  (%d launches)" % (ds[len(ds) // 2], ds[0], len(ds)))
PY

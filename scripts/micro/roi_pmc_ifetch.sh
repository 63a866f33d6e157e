#!/bin/bash
# instruction-fetch counters of a RoI kernel through the torch-free harness.  Usage: roi_pmc_ifetch.sh <kernel substring> [settings]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
PAT=$1; shift
BIN=${ROI_BIN:-$R/scripts/micro/_bin/roi_micro}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp3
timeout 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/rp3 -o p3 -- bash -c "cd $R && $BIN $*" > /tmp/rp3.log 2>&1; echo "rc=$?"
tail -3 /tmp/rp3.log
cd "$R"
python - "$PAT" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
for f in glob.glob("/tmp/rp3/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
    for k, (v, ids) in sorted(acc.items()):
        print("%-28s per launch %14.1f   (%d launches)" % (k, v / max(len(ids), 1), len(ids)))
PY

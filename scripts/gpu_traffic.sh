#!/bin/bash
# HBM traffic of the hot kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slot limits), kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/gpurun_out/traffic_$c" -o t -- python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/traffic_$c.log" 2>&1; echo "$c rc=$?"
done
cd "$R"; python - <<'PY'
import csv, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/traffic_%s/t_counter_collection.csv" % c)):
        if r["Counter_Name"] == c:
            name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70]
            key = name + "|grid=" + r["Grid_Size"]
            acc[key].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[c + "_KB_per_launch"] = sum(v) / len(v)
json.dump(out, open("gpurun_out/r01_hbm_traffic.json", "w"), indent=1, sort_keys=True)
for k in sorted(out):
    print(k[:90], out[k])
PY

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
conv_total = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
conv_launches = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
roi_total = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/traffic_%s/t_counter_collection.csv" % c)):
        if r["Counter_Name"] == c:
            name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70]
            key = name + "|grid=" + r["Grid_Size"]
            acc[key].append(float(r["Counter_Value"]))
            if "conv_mfma_f32_kernel<3" in name:   # the 14 3x3 launches per image the roofline prices (not the 1x1 heads)
                conv_total[c] += float(r["Counter_Value"]); conv_launches[c] += 1
            if "roi_pool_planes_kernel" in name:
                roi_total[c].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[c + "_KB_per_launch"] = sum(v) / len(v)
        out[k]["launches"] = len(v)
# counters are in KB; FETCH_SIZE is NOT doubled here: the conv and RoI kernels issue 4-byte-per-lane loads, for which the counter
# matches known byte counts 1:1 (conv1_2 reads its 153.6 MB input once plus halos); the x2 gfx950 correction of
# the guide applies to 16-byte-per-lane streaming reads only.
out["_summary"] = {
    "command": "python bench.py --steps 3 --warmup 2 --no-cpu-baseline (two passes: --pmc FETCH_SIZE, --pmc WRITE_SIZE, --kernel-trace only)",
    "conv_mfma_f32_kernel": {"launches_counted": conv_launches["FETCH_SIZE"],
                             "hbm_bytes_per_launch": 1024.0 * (conv_total["FETCH_SIZE"] / max(1, conv_launches["FETCH_SIZE"]) + conv_total["WRITE_SIZE"] / max(1, conv_launches["WRITE_SIZE"])),
                             "fetch_bytes_per_launch": 1024.0 * conv_total["FETCH_SIZE"] / max(1, conv_launches["FETCH_SIZE"]),
                             "write_bytes_per_launch": 1024.0 * conv_total["WRITE_SIZE"] / max(1, conv_launches["WRITE_SIZE"])},
    "roi_pool_planes_kernel": {"fetch_bytes_per_launch": 1024.0 * sum(roi_total["FETCH_SIZE"]) / max(1, len(roi_total["FETCH_SIZE"])),
                               "write_bytes_per_launch": 1024.0 * sum(roi_total["WRITE_SIZE"]) / max(1, len(roi_total["WRITE_SIZE"]))}}
json.dump(out, open("gpurun_out/r01_hbm_traffic.json", "w"), indent=1, sort_keys=True)
for k in sorted(out):
    print(k[:90], out[k])
import shutil; shutil.copy("gpurun_out/r01_hbm_traffic.json", "gpurun_out/r01_hbm_traffic_pmc.json")
PY

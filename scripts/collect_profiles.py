#!/usr/bin/env python
"""Copy the evidence of scripts/r02_gpu_final.sh from gpurun_out/<tag>/ into profiles/r02_* (tracked)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag="r02z"):
    O, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    for n in ("r02_bench.json", "r02_bench_default_steps.json", "r02_bench_bf16.json", "r02_bench_f32s.json", "r02_bench_train.json",
              "r02_bench_train_f32s.json", "r02_hbm_traffic_pmc.json", "r02_hbm_traffic_pmc_bf16.json", "r02_hbm_traffic_pmc_f32s.json"):
        shutil.copy(os.path.join(O, n), os.path.join(P, n))
    for n in ("r02_bench_2rank_gloo_infer.json", "r02_bench_2rank_gloo_train.json"):          # drop gloo's log lines
        line = [l for l in open(os.path.join(O, n)) if l.startswith('{"metric"')][-1]
        open(os.path.join(P, n), "w").write(line)
    shutil.copy(os.path.join(O, "parity_reports.txt"), os.path.join(P, "r02_parity_reports.txt"))
    for src, dst in (("prof/r02_kernel_stats.csv", "r02_kernel_stats.csv"), ("prof_bf16/r02_bf16_kernel_stats.csv", "r02_bf16_kernel_stats.csv"),
                     ("prof_f32s/r02_f32s_kernel_stats.csv", "r02_f32s_kernel_stats.csv"), ("prof_train/r02_train_kernel_stats.csv", "r02_train_kernel_stats.csv"),
                     ("prop/prop_kernel_stats.csv", "r02_proposals_kernel_stats.csv")):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))
    summ = {}
    for d, f in (("rpmc1", "p1_counter_collection.csv"), ("rpmc2", "p2_counter_collection.csv")):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(O, d, f))):
            if "roi_pool_cells_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            summ[k] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
    summ["_note"] = ("roi_pool_cells_kernel (fp32 and bf16-output launches of scripts/roi_bench.py), rocprofv3 --pmc, two passes "
                     "(scripts/r02_gpu_final.sh); per-launch means summed over all SEs/CUs")
    json.dump(summ, open(os.path.join(P, "r02_roi_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    for n in ("r02_bench", "r02_bench_default_steps", "r02_bench_bf16", "r02_bench_f32s", "r02_bench_train", "r02_bench_train_f32s"):
        d = json.load(open(os.path.join(P, n + ".json")))
        print(n, round(d["value"], 1), round(d["ms_per_step"], 4), (d.get("roofline") or {}).get("frac"), (d.get("nms_roi") or {}).get("proposals_nms_us"),
              (d.get("nms_roi") or {}).get("roi_pool_us"), (d.get("parity") or {}).get("ok"), (d.get("f32_split_products") or {}).get("value"),
              d.get("ms_per_step_without_proposal_layer"))


if __name__ == "__main__":
    main(*sys.argv[1:])

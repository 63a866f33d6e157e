#!/usr/bin/env python
"""Copy the evidence of scripts/r04_gpu_final.sh from gpurun_out/<tag>/ into profiles/r04_* (tracked)."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json_line(path):
    lines = [l for l in open(path) if l.startswith('{"metric"')]
    return lines[-1] if lines else None


def main(tag="r04z"):
    O, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    for n in ("r04_bench", "r04_bench_bf16", "r04_bench_f32s", "r04_bench_train", "r04_bench_train_f32s", "r04_bench_nccl_w1_train",
              "r04_bench_2rank_gloo_train", "r04_bench_2rank_gloo_infer", "r04_bench_train_rcnn_device", "r04_bench_train_rcnn_numpy", "r04_bench_bf16_pair1", "r04_bench_bf16_pair0"):
        src = os.path.join(O, n + ".json")
        if os.path.exists(src):
            line = last_json_line(src)
            if line:
                open(os.path.join(P, n + ".json"), "w").write(line)
    for n in ("r04_hbm_traffic_pmc.json", "r04_hbm_traffic_pmc_bf16.json", "r04_mfma_pmc_summary.json", "r04_roi_pmc.txt", "r04_store_micro.txt",
              "r04_roi_micro.txt", "r04_conv_bf16_micro.txt", "r04_conv_f32_micro.txt", "r04_wgrad_micro.txt", "r04_mfma_filler_micro.txt", "r04_dma_align_micro.txt", "r04_conv_pair_micro.txt", "r04_mfma_peak_micro.txt"):
        if os.path.exists(os.path.join(O, n)):
            shutil.copy(os.path.join(O, n), os.path.join(P, n))
    if os.path.exists(os.path.join(O, "parity_reports.txt")):
        shutil.copy(os.path.join(O, "parity_reports.txt"), os.path.join(P, "r04_parity_reports.txt"))
    for d, dst in (("prof", "r04_kernel_stats.csv"), ("prof_bf16", "r04_bf16_kernel_stats.csv"), ("prof_train", "r04_train_kernel_stats.csv")):
        hits = glob.glob(os.path.join(O, d, "**", "*kernel_stats.csv"), recursive=True)
        if hits:
            shutil.copy(hits[0], os.path.join(P, dst))
    for n in ("r04_bench", "r04_bench_bf16", "r04_bench_f32s", "r04_bench_train", "r04_bench_train_f32s", "r04_bench_nccl_w1_train", "r04_bench_2rank_gloo_train"):
        f = os.path.join(P, n + ".json")
        if not os.path.exists(f):
            continue
        d = json.load(open(f))
        nr = d.get("nms_roi") or {}
        print(n, round(d["value"], 1), round(d["ms_per_step"], 4), (d.get("roofline") or {}).get("frac"), nr.get("proposals_nms_us"), nr.get("roi_pool_us"),
              nr.get("roi_pool_frac_of_hbm_peak"), (d.get("parity") or {}).get("ok"), (d.get("f32_split_products") or {}).get("value"),
              (d.get("bf16_config3") or {}).get("value"), (d.get("bf16_config3") or {}).get("frac_of_bf16_mfma_peak"))


if __name__ == "__main__":
    main(*sys.argv[1:])

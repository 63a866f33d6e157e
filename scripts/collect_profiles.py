#!/usr/bin/env python
"""Copy the evidence of scripts/gpu_evidence.sh (+ the round's final script) from gpurun_out/<tag>/ into profiles/<prefix>_* (tracked).
Usage: collect_profiles.py <tag> [prefix]   (default r06z r06)"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json_line(path):
    lines = [l for l in open(path) if l.startswith('{"metric"')]
    return lines[-1] if lines else None


def main(tag="r06z", R="r06"):
    O, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    for n in (R + "_bench", R + "_bench_bf16", R + "_bench_f32s", R + "_bench_train", R + "_bench_train_f32s", R + "_bench_nccl_w1_train",
              R + "_bench_2rank_gloo_train", R + "_bench_2rank_gloo_infer", R + "_bench_train_rcnn_device", R + "_bench_train_rcnn_numpy", R + "_bench_train_rcnn_f32s", R + "_bench_train_rcnn_allrows_late", R + "_bench_bf16_pair1", R + "_bench_bf16_pair0", R + "_bench_f16", R + "_bench_resnet101"):
        src = os.path.join(O, n + ".json")
        if os.path.exists(src):
            line = last_json_line(src)
            if line:
                open(os.path.join(P, n + ".json"), "w").write(line)
    for n in (R + "_hbm_traffic_pmc.json", R + "_hbm_traffic_pmc_bf16.json", R + "_mfma_pmc_summary.json", R + "_roi_pmc.txt", R + "_store_micro.txt",
              R + "_roi_micro.txt", R + "_conv_bf16_micro.txt", R + "_conv_f32_micro.txt", R + "_wgrad_micro.txt", R + "_mfma_filler_micro.txt", R + "_dma_align_micro.txt", R + "_conv_pair_micro.txt", R + "_mfma_peak_micro.txt", R + "_roi_bwd_pmc.txt", R + "_bench_power.txt", R + "_two_streams_probe.txt", R + "_mfma_pmc_summary_f32.json", R + "_linear_bf16_micro_final.txt"):
        if os.path.exists(os.path.join(O, n)):
            shutil.copy(os.path.join(O, n), os.path.join(P, n))
    if os.path.exists(os.path.join(O, "pytest_gpu.log")):          # the GPU suite's own last lines (counts) next to the PARITY reports
        tail = [l for l in open(os.path.join(O, "pytest_gpu.log")) if (" passed" in l or " failed" in l or l.startswith(("FAILED", "ERROR")))]
        open(os.path.join(P, R + "_pytest_gpu_summary.txt"), "w").write("".join(tail[-12:]))
    if os.path.exists(os.path.join(O, "parity_reports.txt")):
        shutil.copy(os.path.join(O, "parity_reports.txt"), os.path.join(P, R + "_parity_reports.txt"))
    for d, dst in (("prof", R + "_kernel_stats.csv"), ("prof_bf16", R + "_bf16_kernel_stats.csv"), ("prof_train", R + "_train_kernel_stats.csv"), ("prof_two", R + "_bf16_two_in_flight_kernel_stats.csv"),
                   ("prof_rcnn", R + "_train_rcnn_kernel_stats.csv"), ("prof_f16", R + "_f16_kernel_stats.csv")):
        hits = glob.glob(os.path.join(O, d, "**", "*kernel_stats.csv"), recursive=True)
        if hits:
            shutil.copy(hits[0], os.path.join(P, dst))
    for n in (R + "_bench", R + "_bench_bf16", R + "_bench_f32s", R + "_bench_train", R + "_bench_train_f32s", R + "_bench_nccl_w1_train", R + "_bench_2rank_gloo_train"):
        f = os.path.join(P, n + ".json")
        if not os.path.exists(f):
            continue
        d = json.load(open(f))
        nr = d.get("nms_roi") or {}
        print(n, round(d["value"], 1), round(d["ms_per_step"], 4), (d.get("roofline") or {}).get("frac"), nr.get("proposals_nms_us"), nr.get("roi_pool_us"),
              nr.get("roi_pool_frac_of_hbm_peak"), (d.get("parity") or {}).get("ok"), (d.get("f32_split_products") or {}).get("value"),
              (d.get("bf16_config3") or {}).get("value"), (d.get("bf16_config3") or {}).get("frac_of_bf16_mfma_peak"), (d.get("f16_config3") or {}).get("value"))


if __name__ == "__main__":
    main(*sys.argv[1:])

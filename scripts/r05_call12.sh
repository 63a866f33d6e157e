#!/bin/bash
# Round 5, GPU call: the bench's secondary lines -- input inside the timed region (with_feed) and two images in flight per GPU -- on the fp32 and bf16 lines.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/r05r; mkdir -p $O
for dt in f32 bf16; do
  timeout 600 python bench.py --dtype $dt --no-split-variant --no-cpu-baseline > $O/bench_$dt.json 2>> $O/bench.err; echo "$dt rc=$?"
  python - "$O/bench_$dt.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
print("  value %.1f img/s (%.4f ms)" % (d["value"], d["ms_per_step"]))
print("  with_feed", {k: v for k, v in (d.get("with_feed") or {}).items() if k != "what"})
print("  two_images_in_flight", {k: v for k, v in (d.get("two_images_in_flight") or {}).items() if k not in ("what", "stream_set_probe_img_s")})
b = d.get("bf16_config3") or {}
if b:
    print("  bf16_config3 value %.1f" % b["value"], "with_feed", (b.get("with_feed") or {}).get("img_s_with_feed"), "two", {k: v for k, v in (b.get("two_images_in_flight") or {}).items() if k != "what"})
PY
done
tail -3 $O/bench.err
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -k "forwards_in_flight or captured" 2>&1 | tail -4

#!/bin/bash
# Package power and shader clock while the bench's timed regions run (hipGraph replays of the whole forward): the fp32 contract line and the bf16 line.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=${O:-gpurun_out/power}; mkdir -p $O
poll() { for i in $(seq 1 $1); do echo "t=$(date +%s.%N) $(/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Current Socket' | tr -s ' ' | tr '\n' '|')"; sleep 0.1; done; }
for dt in f32 bf16; do
  steps=2500; [ $dt = bf16 ] && steps=12000
  poll 150 > $O/smi_$dt.txt 2>&1 &
  P=$!
  echo "start $(date +%s.%N)" > $O/marks_$dt.txt
  timeout 300 python bench.py --dtype $dt --steps $steps --warmup 5 --no-cpu-baseline --no-split-variant --no-bf16-variant --no-stage-events --no-feed-variant --no-two-streams-variant > $O/bench_$dt.json 2> $O/err_$dt.txt
  echo "end $(date +%s.%N)" >> $O/marks_$dt.txt
  wait $P
  python - "$O" "$dt" <<'PY'
import json, re, sys
O, dt = sys.argv[1], sys.argv[2]
d = json.loads([l for l in open("%s/bench_%s.json" % (O, dt)) if l.startswith('{"metric"')][-1])
rows = []
for l in open("%s/smi_%s.txt" % (O, dt)):
    m = re.search(r't=([\d.]+).*?\((\d+)Mhz\).*?Power \(W\): ([\d.]+)', l)
    if m: rows.append((float(m.group(1)), int(m.group(2)), float(m.group(3))))
end = float(open("%s/marks_%s.txt" % (O, dt)).read().split("end ")[1])
timed = d["steps"] * d["ms_per_step"] / 1e3
sel = [(c, p) for t, c, p in rows if end - timed - 0.3 < t < end - 0.5]
print(dt, round(d["value"], 1), "img/s; timed region", round(timed, 2), "s; samples inside", len(sel), "sclk min/median/max", min(c for c, _ in sel), sorted(c for c, _ in sel)[len(sel) // 2], max(c for c, _ in sel), "MHz; power min/median/max", min(p for _, p in sel), sorted(p for _, p in sel)[len(sel) // 2], max(p for _, p in sel), "W")
PY
done

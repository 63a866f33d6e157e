#!/usr/bin/env python
"""Static check for dependent memory round trips the compiler put into a kernel: compiles every csrc/*.hip to gfx950 assembly
(hipcc -S, no GPU needed) and lists, per kernel, the global / buffer loads, the `s_waitcnt vmcnt(0)` instructions and the stores.
A kernel whose vmcnt(0) count is of the order of its load count is waiting for its loads one at a time (DESIGN.md section 3.10:
predicated loads inside unrolled load - use - store loops); the listing under /tmp/isa/<file>.s shows where.

    python scripts/isa_wait_scan.py [min_waits]        # default: kernels with >= 6 full waits
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chainer-faster-rcnn_amd", "csrc")


def main(min_waits=6):
    out_dir = "/tmp/isa"
    os.makedirs(out_dir, exist_ok=True)
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        asm = os.path.join(out_dir, f[:-4] + ".s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                        os.path.join(CSRC, f), "-o", asm], check=True, stderr=subprocess.DEVNULL)
        name, stats = None, {}
        for ln in open(asm):
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                name = m.group(1)
                stats[name] = [0, 0, 0]
            elif name is None:
                continue
            elif ln.startswith(".Lfunc_end"):
                name = None
            elif re.search(r"\b(global_load|buffer_load)", ln) and " lds" not in ln:
                stats[name][0] += 1
            elif "s_waitcnt vmcnt(0)" in ln:
                stats[name][1] += 1
            elif re.search(r"\b(global_store|buffer_store)", ln):
                stats[name][2] += 1
        for k, (loads, waits, stores) in stats.items():
            if waits >= min_waits:
                demangled = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
                demangled = demangled.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                print("%-14s loads %4d  vmcnt(0) %4d  stores %4d  %s" % (f, loads, waits, stores, demangled[:90]))


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])

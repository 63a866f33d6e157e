"""Build libfrcnn_hip.so in-tree: hipcc --offload-arch=gfx950, one object per .hip, then link."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT_SO = os.path.join(PKG, "libfrcnn_hip.so")
OBJ = os.path.join(HERE, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-I", os.path.join(ROOT, "include"), "-I", HERE]
if os.environ.get("FRCNN_TIMING_ABLATIONS") == "1":      # tuning builds only: adds kernels that skip work (wrong results) for the sweeps
    FLAGS.append("-DFRCNN_TIMING_ABLATIONS")


def build(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    hdrs = glob.glob(os.path.join(HERE, "*.h")) + [os.path.join(ROOT, "include", "frcnn_hip.h")]
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    os.makedirs(OBJ, exist_ok=True)
    procs, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_hdr):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        try:
            rc = p.wait(timeout=900)          # a pathological unroll must fail loudly, not hang the build
        except subprocess.TimeoutExpired:
            p.kill()
            raise RuntimeError("hipcc timed out on %s" % s)
        if rc != 0:
            raise RuntimeError("hipcc failed on %s" % s)
    if procs or not os.path.exists(OUT_SO):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT_SO] + objs)
    return OUT_SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""Compile the UNMODIFIED kernel sources of chainer-faster-rcnn_amd/csrc/ for the host, with
tests/hipemu/hip/hip_runtime.h shadowing the real HIP header -> tests/hipemu/_build/libfrcnn_emu.so.
Test infrastructure only (see the header's banner)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "chainer-faster-rcnn_amd", "csrc")
OUT = os.path.join(HERE, "_build")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build(force=False, sources=None):
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, "libfrcnn_emu.so")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) if sources is None else sources
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "shadow", "*.h")) + [os.path.join(HERE, "hip", "hip_runtime.h"),
                                                          os.path.join(ROOT, "include", "frcnn_hip.h")]
    if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps):
        return so
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + ".o")
        objs.append(o)
        procs.append(subprocess.Popen([CLANG if os.path.exists(CLANG) else "g++", "-x", "c++", "-std=c++17", "-O1", "-g",
                                       "-ffp-contract=off", "-fPIC", "-Wno-unused-function", "-Wno-unused-value", "-DFRCNN_TUNING_FORMS",     # the emulator carries the research forms too (logic tests)
                                       
                                       "-I", os.path.join(HERE, "shadow"), "-I", HERE, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                                       "-c", s, "-o", o]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipemu build failed")
    subprocess.check_call([CLANG if os.path.exists(CLANG) else "g++", "-shared", "-o", so] + objs)
    return so


if __name__ == "__main__":
    print(build(force=True))

"""Build the reference's own native CPU code into oracle/_ref/ (TEST INFRASTRUCTURE ONLY).

The reference ships two Cython sources on the hot path:
  /root/reference/models/cpu_nms.pyx  (greedy IoU NMS, cpu_nms.pyx:18-69)
  /root/reference/models/bbox.pyx     (dense IoU matrix,  bbox.pyx:16-56)
They are compiled here *from where they lie* under /root/reference; nothing is copied
into the tracked tree.  Outputs (generated .pyx/.c/.so) go only into oracle/_ref/,
which is git-ignored but travels to the GPU box with the gpurun snapshot.

cpu_nms.pyx does not cythonize under NumPy 2 / Cython 3 because of the removed aliases
`np.int_t` / `np.int`; a 2-token textual patch (np.int_t -> np.intp_t,
dtype=np.int -> dtype=np.intp) is applied to the *generated copy* in oracle/_ref/.
The arithmetic is untouched.  bbox.pyx cythonizes unmodified but needs `np.float`
restored at import time (oracle/ref_harness.py does that).

If /root/reference is absent (GPU box) this is a no-op: the prebuilt .so files are used.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("FRCNN_REFERENCE", "/root/reference")


def _ext_suffix():
    return sysconfig.get_config_var("EXT_SUFFIX")


def built():
    return all(os.path.exists(os.path.join(OUT, n + _ext_suffix())) for n in ("cpu_nms", "bbox"))


def build(force=False):
    if not os.path.isdir(os.path.join(REF, "models")):
        return built()
    if built() and not force:
        return True
    import numpy as np
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(REF, "models", "cpu_nms.pyx")).read()
    src = src.replace("np.int_t", "np.intp_t").replace("dtype=np.int)", "dtype=np.intp)")
    open(os.path.join(OUT, "cpu_nms.pyx"), "w").write(src)
    open(os.path.join(OUT, "bbox.pyx"), "w").write(open(os.path.join(REF, "models", "bbox.pyx")).read())
    inc = ["-I" + sysconfig.get_paths()["include"], "-I" + np.get_include()]
    for name in ("cpu_nms", "bbox"):
        pyx = os.path.join(OUT, name + ".pyx")
        c = os.path.join(OUT, name + ".c")
        subprocess.check_call([sys.executable, "-m", "cython", "-3", pyx, "-o", c])
        so = os.path.join(OUT, name + _ext_suffix())
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-w",
                               "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION", c, "-o", so] + inc)
        os.remove(pyx)   # keep only the binaries: no reference source text stays in the tree
        os.remove(c)
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("oracle/_ref built:", ok)

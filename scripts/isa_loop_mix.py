"""Instruction mix of the basic blocks that hold MFMAs, per kernel, from the gfx950 assembly hipcc leaves with -save-temps.
Usage: python scripts/isa_loop_mix.py <file.hip> <kernel-name-regex>     (compiles into /tmp/isa, prints one line per MFMA block)
DESIGN 3.11: on a CDNA4 SIMD every non-MFMA instruction beyond one per MFMA is paid in matrix-pipe time, so this count -- not
occupancy -- is what a kernel's main loop is judged by before it is ever timed."""
import collections
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chainer-faster-rcnn_amd", "csrc")


def assemble(src, out="/tmp/isa"):
    os.makedirs(out, exist_ok=True)
    base = os.path.splitext(os.path.basename(src))[0]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-w", "-I", os.path.join(ROOT, "include"),
                    "-I", CSRC, "-c", src, "-o", os.path.join(out, base + ".o"), "-save-temps=obj"], check=True, cwd=out)
    return glob.glob(os.path.join(out, base + "-hip-amdgcn*.s"))[0]


def kernels(asm):
    s = open(asm).read()
    for m in re.finditer(r"^(\S+):\s*; @\1\s*$", s, re.M):
        end = s.index(".end_amdhsa_kernel", m.end()) if ".amdhsa_kernel " + m.group(1) in s else None
        if end is None:
            continue
        body = s[m.end():s.index("s_endpgm", m.end())]
        meta = s[s.index(".amdhsa_kernel " + m.group(1)):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        yield m.group(1), body, meta


def mix(body):
    blocks, cur = [], ["entry", []]
    for l in body.split("\n"):
        l = l.strip()
        if re.match(r"^\.LBB\S+:", l):
            blocks.append(cur)
            cur = [l.split(":")[0], []]
        elif l and not l.startswith(";") and not l.startswith("."):
            cur[1].append(l.split()[0])
    blocks.append(cur)
    for lab, ins in blocks:
        c = collections.Counter(ins)
        n = sum(v for k, v in c.items() if k.startswith("v_mfma"))
        if n:
            yield lab, len(ins), n, c


def main():
    asm = sys.argv[1] if sys.argv[1].endswith(".s") else assemble(os.path.abspath(sys.argv[1]))
    pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    for name, body, meta in kernels(asm):
        if not pat.search(name):
            continue
        g = lambda k: re.search(k + r"\s+(\S+)", meta).group(1)
        print("%s\n  vgpr %s (accum offset %s)  sgpr %s  lds %s  scratch %s" % (name, g(".amdhsa_next_free_vgpr"), g(".amdhsa_accum_offset"), g(".amdhsa_next_free_sgpr"),
                                                                             g(".amdhsa_group_segment_fixed_size"), g(".amdhsa_private_segment_fixed_size")))
        for lab, n, nm, c in mix(body):
            keys = ["ds_read_b128", "buffer_load_dwordx4", "s_waitcnt", "s_barrier", "s_nop", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_mov_b32_e32", "s_mov_b32"]
            rest = n - nm - sum(c.get(k, 0) for k in keys)
            print("  %-10s %4d instr: %3d mfma | %s | other %d" % (lab, n, nm, " ".join("%s %d" % (k.replace("buffer_load_dwordx4", "dma").replace("_b32", "").replace("_e32", ""), c.get(k, 0)) for k in keys), rest))


if __name__ == "__main__":
    main()

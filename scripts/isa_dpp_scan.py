"""CPU-only: every DPP read in a gfx950 listing must sit at least two wait states behind the last VALU write of the VGPR it reads
(the ISA's "VALU writes VGPR -> DPP reads that VGPR" hazard; an `s_nop N` counts N + 1, any other instruction 1).  The compiler's hazard
recognizer covers its own DPP instructions but sees nothing inside an inline-asm string -- neither a DPP read nor a VALU write -- so the
scan looks at the final listing, whoever produced the instructions.  Usage: isa_dpp_scan.py file.s [...]; tests/test_isa_waits.py calls
`scan()`."""
import re
import sys

_DPP = re.compile(r"\b(quad_perm|row_shl|row_shr|row_ror|wave_shl|wave_shr|wave_rol|wave_ror|row_mirror|row_half_mirror|row_bcast|row_newbcast)\b")
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _regs(tok):
    out = set()
    for m in _REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(path):
    """-> list of (function, line number, instruction, register, wait states found)"""
    bad, fn = [], None
    window = []                       # (written VGPRs, wait states this instruction contributes), most recent last
    for no, raw in enumerate(open(path), 1):
        ln = raw.split(";")[0].strip()
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            fn, window = m.group(1), []
            continue
        if not ln or ln.startswith(".") or ln.endswith(":") or fn is None:
            if ln.endswith(":"):
                window = []           # a label: control flow may join here; the look-back restarts (branches cost far more than two states)
            continue
        parts = ln.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        if op == "s_nop":
            window.append((set(), int(args.strip() or 0) + 1))
            continue
        toks = [t.strip() for t in args.split(",")]
        if _DPP.search(ln) and op.startswith("v_"):
            # sources: everything after the destination; the DPP lane select applies to src0
            src0 = _regs(toks[1]) if len(toks) > 1 else set()
            states = 0
            for written, ws in reversed(window):
                hit = written & src0
                if hit:
                    if states < 2:
                        bad.append((fn, no, ln, sorted(hit)[0], states))
                    break
                states += ws
                if states >= 2:
                    break
        written = _regs(toks[0]) if op.startswith("v_") and toks else set()
        window.append((written, 1))
        if len(window) > 8:
            window.pop(0)
    return bad


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        for fn, no, ln, reg, st in scan(p):
            print("%s:%d %s: v%d read through DPP %d wait state(s) after its write: %s" % (p, no, fn[:60], reg, st, ln))
            rc = 1
    sys.exit(rc)

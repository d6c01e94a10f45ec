#!/usr/bin/env python3
"""Scan the gfx950 code of kernels_m2l_rot.o for the hazard the compiler cannot see inside the inline asm: a VALU write of a
VGPR that a DPP instruction reads as its DPP source (src0) fewer than two wait states later.
usage: tools/check_dpp_hazard.py kernels_m2l_rot.o   -> per kernel: DPP FMAs, hazards (exit 1 if any)"""
import re
import subprocess
import sys
import tempfile
import os

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, "k.o")
        with open(obj, "rb") as f, open(tmp, "wb") as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, "--offloading", "k.o"], cwd=td, check=True, capture_output=True)
        co = [x for x in os.listdir(td) if "gfx950" in x]
        if not co:
            raise SystemExit("no gfx950 code object in " + obj)
        return subprocess.run([OBJDUMP, "-d", co[0]], cwd=td, check=True, capture_output=True, text=True).stdout


def regs(tok):
    """VGPR numbers named by an operand token: v5, v[4:5]"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main():
    text = disassemble(sys.argv[1])
    bad_total = 0
    name = None
    window = []                                   # (wait states this instruction provides, VGPRs it writes as a VALU)
    stats = {}
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            name = m.group(1)
            window = []
            stats[name] = [0, 0]
            continue
        ins = line.split("//")[0].strip()
        if not ins or name is None:
            continue
        op, _, rest = ins.partition(" ")
        ops = [t.strip() for t in rest.split(",")]
        if op.endswith("_dpp"):
            stats[name][0] += 1
            src0 = regs(ops[1].split()[0]) if len(ops) > 1 else set()
            ws = 0
            for states, written in reversed(window):
                if ws >= 2:
                    break
                if written & src0:
                    stats[name][1] += 1
                    bad_total += 1
                    break
                ws += states
        if op == "s_nop":
            window.append((int(ops[0], 0) + 1, set()))
        elif op.startswith("v_") and not op.startswith("v_cmp") and ops:
            window.append((1, regs(ops[0].split()[0])))
        else:
            window.append((1, set()))
        window = window[-4:]
    for k, (n, bad) in stats.items():
        if n:
            m = re.search(r"kernelILi(\d+)E", k)
            print("p=%-3s dpp %5d  hazards %d" % (m.group(1) if m else k, n, bad))
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()

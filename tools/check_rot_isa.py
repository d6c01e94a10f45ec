#!/usr/bin/env python3
"""Checks on the gfx950 code of kernels_m2l_rot.o that the compiler cannot make, because the instructions sit in inline asm:

1. DPP hazard: a VALU write of a VGPR that a DPP instruction reads as its DPP source (src0) fewer than two wait states later.
2. Loads issued by hand: the kernel issues global loads in asm long before it waits for them (s_waitcnt vmcnt(N), in-order
   return).  To the compiler such an asm has produced its result when it is over, so it is free to copy or spill the
   destination at once -- and would copy what the register held BEFORE the load.  The check follows every vector-memory load
   from its issue to the s_waitcnt that covers it, along every path of the kernel's control flow, and reports any instruction
   in between that touches its destination registers.

usage: tools/check_rot_isa.py kernels_m2l_rot.o    -> a line per kernel; exit 1 on any finding
       tools/check_rot_isa.py --text listing.s      the same on a disassembly listing (llvm-objdump -d text; tests feed doctored ones)
       tools/check_rot_isa.py --nop-mask report     reads a report printed by this script and prints the FMMBEM_ROT_NOP_ORDERS mask
                                                    (bit p - 1) that cures it: the orders with DPP hazards -- or 0 when the report
                                                    also holds early touches of loads in flight, which no wait state cures.
The build (csrc/Makefile ROTBUILD) rebuilds an object ONCE with that mask and fails only if the second object is red too."""
import os
import re
import subprocess
import sys
import tempfile

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
    """registers named by an operand token: {('v', 5)}, v[4:5], a[0:3]"""
    tok = tok.strip().split()[0] if tok.strip() else ""
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    return set()


class Ins:
    __slots__ = ("addr", "op", "ops", "text")

    def __init__(self, addr, op, ops, text):
        self.addr, self.op, self.ops, self.text = addr, op, ops, text


def parse(text):
    kernels, cur = {}, None
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None or "//" not in line:
            continue
        code, comment = line.split("//", 1)
        code = code.strip()
        m = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if not code or not m:
            continue
        op, _, rest = code.partition(" ")
        cur.append(Ins(int(m.group(1), 16), op, [t for t in rest.split(",")] if rest.strip() else [], code))
    return kernels


def is_vmem(op):
    return op.startswith(("global_load", "global_store", "global_atomic", "scratch_load", "scratch_store", "buffer_load",
                          "buffer_store", "buffer_atomic", "flat_load", "flat_store", "flat_atomic"))


def check_dpp(ins):
    bad, window = [], []
    for i in ins:
        if i.op.endswith("_dpp"):
            src0 = {r for r in regs(i.ops[1]) if r[0] == "v"} if len(i.ops) > 1 else set()
            ws = 0
            for states, written in reversed(window):
                if ws >= 2:
                    break
                if written & src0:
                    bad.append(i)
                    break
                ws += states
        if i.op == "s_nop":
            window.append((int(i.ops[0], 0) + 1, set()))
        elif i.op.startswith("v_") and not i.op.startswith("v_cmp") and i.ops:
            window.append((1, regs(i.ops[0])))
        else:
            window.append((1, set()))
        window = window[-4:]
    return bad


def step(i, pending, found):
    """advance the in-flight list over instruction i; report touches of registers still in flight"""
    touched = set()
    for t in i.ops:
        touched |= regs(t)
    if is_vmem(i.op) and "load" in i.op and i.ops:
        touched -= regs(i.ops[0])                 # its own destination (an in-order queue makes write-after-write safe)
    for dest in pending:
        if dest & touched:
            found.append(i)
            break
    if i.op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", i.text)
        if m:
            n = int(m.group(1))
            del pending[:max(0, len(pending) - n)]
    elif is_vmem(i.op):
        pending.append(regs(i.ops[0]) if ("load" in i.op and i.ops) else set())
    # what cannot matter any more: the oldest entries while they carry no register (stores; they complete first), and
    # anything beyond the 64 the counter can hold
    del pending[:max(0, len(pending) - 64)]
    while pending and not pending[0]:
        del pending[0]


def branch_target(i):
    off = int(i.ops[0].strip(), 0)
    off = off - 65536 if off >= 32768 else off
    return i.addr + 4 + 4 * off


def check_loads(ins):
    """walk every path of the kernel's control flow with the list of loads in flight (states met before are not walked again)"""
    index = {i.addr: k for k, i in enumerate(ins)}
    is_branch = [i.op.startswith(("s_cbranch", "s_branch")) and bool(i.ops) for i in ins]
    starts = {0}
    for k, i in enumerate(ins):
        if is_branch[k]:
            starts.add(k + 1)
            if branch_target(i) in index:
                starts.add(index[branch_target(i)])
    found, seen, stack = [], set(), [(0, ())]
    while stack:
        k, state = stack.pop()
        pending = [set(x) for x in state]
        while k < len(ins):
            if k in starts:
                key = (k, tuple(frozenset(x) for x in pending))
                if key in seen:
                    break
                seen.add(key)
            i = ins[k]
            step(i, pending, found)
            if i.op == "s_endpgm":
                break
            if is_branch[k]:
                t = index.get(branch_target(i))
                if i.op == "s_branch":
                    if t is None:
                        break
                    k = t
                    continue
                if t is not None:
                    stack.append((t, tuple(frozenset(x) for x in pending)))
            k += 1
    uniq = {}
    for i in found:
        uniq.setdefault(i.addr, i)
    return list(uniq.values())


def nop_mask(report):
    mask, incurable = 0, False
    for ln in open(report):
        m = re.match(r"p=(\d+)\s+dpp\s+\d+\s+dpp hazards (\d+)\s+early touches of loads in flight (\d+)", ln)
        if not m:
            continue
        if int(m.group(3)):
            incurable = True
        if int(m.group(2)):
            mask |= 1 << (int(m.group(1)) - 1)
    return 0 if incurable else mask


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--nop-mask":
        print("0x%x" % nop_mask(sys.argv[2]))
        return
    text = open(sys.argv[2]).read() if len(sys.argv) >= 3 and sys.argv[1] == "--text" else disassemble(sys.argv[1])
    kernels = parse(text)
    total = 0
    for name, ins in kernels.items():
        ndpp = sum(1 for i in ins if i.op.endswith("_dpp"))
        if not ndpp:
            continue
        dpp, loads = check_dpp(ins), check_loads(ins)
        m = re.search(r"kernelILi(\d+)E", name)
        print("p=%-3s dpp %5d  dpp hazards %d  early touches of loads in flight %d"
              % (m.group(1) if m else name, ndpp, len(dpp), len(loads)))
        for i in (dpp + loads)[:int(os.environ.get("ROT_ISA_SHOW", "6"))]:
            print("      %x: %s" % (i.addr, i.text))
        total += len(dpp) + len(loads)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()

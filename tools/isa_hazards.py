#!/usr/bin/env python3
"""Data hazards of gfx950 that the hardware does NOT interlock, checked in the disassembly of the built objects.

LLVM's hazard recogniser pads these with s_nop for the instructions it emits itself; it does not look inside inline asm —
neither at an inline-asm instruction as the reader nor as the writer.  csrc/ writes out v_add_f32_dpp (kernels_block.hip),
v_pk_mul_f32 with op_sel and v_pk_fma_f32 ... clamp (pk_common.h) by hand, so the distance between those instructions and
their neighbours is an argument about the schedule, not a guarantee.  This tool turns the argument into a check:

  dpp    a DPP instruction reads (as its DPP operand, src0) a VGPR that a VALU instruction wrote fewer than 2 wait states before
  exec   a DPP instruction follows a VALU write of EXEC (v_cmpx*, v_readfirstlane ... exec) by fewer than 5 wait states
  trans  a non-transcendental VALU instruction reads a VGPR that a transcendental one (v_rsq/rcp/sqrt/exp/log/sin/cos)
         wrote in the instruction right before it (gfx940+: one wait state)

A wait state = one issued instruction; `s_nop N` counts N + 1.  The walk is over the layout order of every function of the
code object; a branch counts as one instruction, so a DPP instruction right at a branch target is checked against the
fall-through path only (the written-out instructions of csrc/ sit in straight-line epilogues).

    python3 tools/isa_hazards.py parallelnbody_amd/csrc/kernels_block.o [more .o ...]      # exit code 1 if anything is found
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TRANS = re.compile(r"^v_(rsq|rcp|sqrt|exp|log|sin|cos)_")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
DPP_CTRL = ("quad_perm", "row_shl", "row_shr", "row_ror", "wave_shl", "wave_shr", "wave_rol", "wave_ror", "row_mirror",
            "row_half_mirror", "row_bcast", "row_newbcast", "row_share", "row_xmask")


def code_object(obj, workdir):
    """The gfx950 code object inside a host .o (hipcc -c) or the file itself if it already is one."""
    head = open(obj, "rb").read(20)
    if head[18:20] == b"\xe0\x00":                               # e_machine = EM_AMDGPU
        return obj
    local = os.path.join(workdir, os.path.basename(obj))
    shutil.copy(obj, local)                                     # llvm-objdump --offloading writes next to its input
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
    for name in sorted(os.listdir(workdir)):
        if name.startswith(os.path.basename(obj) + ".") and "amdgcn" in name:
            return os.path.join(workdir, name)
    raise RuntimeError(f"{obj}: no amdgcn code object inside")


def regs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(line):
    """(mnemonic, [operand strings], modifiers text) of one disassembly line, or None."""
    code = line.split("//")[0].strip()
    if not code or code.endswith(":") or code.startswith("<") or code.startswith("Disassembly") or "file format" in code:
        return None
    parts = code.split(None, 1)
    mnem = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    ops = [o.strip() for o in rest.split(",")]
    return mnem, ops, rest


def vgpr_writes(mnem, ops):
    """VGPRs a VALU instruction writes (its first operand; v_cmp* write SGPRs / VCC / EXEC only)."""
    if not mnem.startswith("v_") or mnem.startswith("v_cmp") or mnem.startswith("v_readlane") or mnem.startswith("v_readfirstlane"):
        return set()
    return regs(ops[0]) if ops else set()


def check(disassembly):
    """[(function, line number, kind, text)] of the hazards found in llvm-objdump -d output."""
    found, func = [], "?"
    window = []                                # [(wait states this instruction provides, vgprs written by VALU, writes EXEC from VALU, is trans, text)]
    for n, line in enumerate(disassembly.splitlines(), 1):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
        if m:
            func, window = m.group(1), []
            continue
        p = parse(line)
        if p is None:
            continue
        mnem, ops, rest = p
        is_dpp = mnem.endswith("_dpp") or any(c in rest for c in DPP_CTRL)
        is_valu = mnem.startswith("v_")
        if is_dpp:
            src0 = regs(ops[1]) if len(ops) > 1 else set()
            dist = 0
            for ws, wr, ex, _tr, text in reversed(window):
                if dist < 2 and wr & src0:
                    found.append((func, n, "dpp", f"{line.strip()}   <- {text.strip()} ({dist} wait state(s) between)"))
                if dist < 5 and ex:
                    found.append((func, n, "exec", f"{line.strip()}   <- {text.strip()} ({dist} wait state(s) between)"))
                dist += ws
                if dist >= 5:
                    break
        if is_valu and not TRANS.match(mnem) and window:
            ws, wr, _ex, tr, text = window[-1]
            read = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            if mnem.startswith("v_cmp") or mnem.startswith("v_readlane") or mnem.startswith("v_readfirstlane"):
                read |= regs(ops[0]) if not ops[0].startswith(("s", "vcc", "exec")) else set()
            if tr and wr & read:
                found.append((func, n, "trans", f"{line.strip()}   <- {text.strip()}"))
        ws = 1
        if mnem == "s_nop":
            ws = int(ops[0], 0) + 1
        writes = vgpr_writes(mnem, ops)
        exec_w = is_valu and (mnem.startswith("v_cmpx") or (ops and ops[0].startswith("exec")))
        window.append((ws, writes, exec_w, bool(TRANS.match(mnem)), line))
        if len(window) > 8:
            window.pop(0)
    return found


def check_object(obj):
    with tempfile.TemporaryDirectory() as d:
        co = code_object(obj, d)
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    n_dpp = sum(1 for ln in dis.splitlines() if "_dpp" in ln.split("//")[0])
    return check(dis), n_dpp, dis.count("\n")


def scratch_in_innermost_loops(obj):
    """{function: (scratch instructions, of them inside an innermost loop)} for the functions of a built object that touch scratch.
    A loop = a backward branch; innermost = no other backward branch's range inside its own.  (Spills parked AROUND a hot loop
    cost a store and a load per trip of the loop around it; spills inside the innermost loops are what a kernel must not have.)"""
    with tempfile.TemporaryDirectory() as d:
        co = code_object(obj, d)
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    return scratch_in_innermost_loops_of(dis)


def scratch_in_innermost_loops_of(dis):
    """The same from llvm-objdump -d text."""
    out, func, insts = {}, None, []

    def close():
        if func is None or not any("scratch_" in t for _a, t in insts):
            return
        loops = []
        for a, t in insts:
            m = re.search(r"s_cbranch_\w+\s+\S+\s+<[^>]*\+0x([0-9a-f]+)>|s_branch\s+\S+\s+<[^>]*\+0x([0-9a-f]+)>", t)
            if m:
                tgt = base + int(m.group(1) or m.group(2), 16)
                if tgt <= a:
                    loops.append((tgt, a))
        inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
        sc = [a for a, t in insts if "scratch_" in t]
        out[func] = (len(sc), sum(1 for a in sc if any(lo <= a <= hi for lo, hi in inner)))

    base = 0
    for line in dis.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line.strip())
        if m:
            close()
            base, func, insts = int(m.group(1), 16), m.group(2), []
            continue
        m = re.search(r"//\s*([0-9A-Fa-f]+):", line)
        if m and func is not None and parse(line) is not None:
            insts.append((int(m.group(1), 16), line.split("//")[0]))
    close()
    return out


def main(argv):
    bad = 0
    for obj in argv:
        found, n_dpp, n_lines = check_object(obj)
        print(f"{obj}: {n_lines} lines, {n_dpp} DPP instructions, {len(found)} hazard(s)")
        for func, n, kind, text in found[:40]:
            print(f"  [{kind}] {func} line {n}: {text}")
        bad += len(found)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

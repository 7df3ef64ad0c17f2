#!/usr/bin/env python3
"""Build-time audit for the stale-register hazard of DESIGN.md 8b: a vector instruction that the compiler placed at the TOP of a
join block, in front of the `s_or_b64 exec, exec, sN` that restores the lanes of the region being left.  Such an instruction runs
under the NARROWED mask of the region (possibly EXEC = 0), so a register copy there (`v_accvgpr_write`, `v_mov`, `v_accvgpr_read`)
leaves the lanes that skipped the region with whatever the register held before -- the neural-ODE adjoint read exactly that.

For every kernel of an object file: basic blocks are taken from `llvm-objdump -d --symbolize-operands`; a block is a JOIN block if
it contains `s_or_b64 exec, exec, ...`; every EXEC-dependent vector instruction in front of that restore is reported
(v_readlane / v_writelane / v_readfirstlane ignore EXEC and are not).  Exit code 1 if a hazard is found in a kernel whose name
matches one of --strict patterns (the kernels whose control flow is genuinely divergent: lane groups smaller than a wavefront).

Reading the report: the structurizer also merges the TAIL of a region with the join block that follows it, so an instruction in
front of the restore may simply belong to the lanes that ran the region (a loop-carried pointer increment, say) -- a candidate
site is a hazard only if the value it writes is needed by the lanes that skipped the region.  The report narrows a review down to
a handful of sites per kernel (register COPIES, `v_accvgpr_*` / `v_mov`, are the ones to look at); the run-time gate that a kernel
does not read lanes it never wrote is tests/test_gpu_poison.py (every register of the chip filled with lane-varying garbage in
front of every kernel: a stale read fails on every run).

usage: python tools/isa_exec_audit.py [--strict SUBSTR ...] file.o [file.o ...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
IGNORES = ("v_readlane", "v_writelane", "v_readfirstlane", "v_nop")


def device_code(obj, tmp):
    dst = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, dst)
    subprocess.run([OBJDUMP, "--offloading", dst], capture_output=True, cwd=tmp)
    for f in os.listdir(tmp):
        if f.startswith(os.path.basename(obj) + ".") and "amdgcn" in f:
            return os.path.join(tmp, f)
    return None


def audit(co):
    txt = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", co], capture_output=True, text=True).stdout.splitlines()
    hazards = {}
    kernel, block = None, []

    def flush():
        if kernel is None:
            return
        for i, ins in enumerate(block):
            if re.match(r"s_or_b64 exec, exec,", ins) or re.match(r"s_or_saveexec_b64", ins):
                for pre in block[:i]:
                    op = pre.split()[0]
                    if op.startswith("v_") and not op.startswith(IGNORES):
                        hazards.setdefault(kernel, []).append(pre)
                break

    for line in txt:
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m and not re.match(r"^L\d+$", m.group(1)):
            flush()
            kernel, block = m.group(1), []
            continue
        if re.match(r"^<L\d+>:", line.strip()) or (m and re.match(r"^L\d+$", m.group(1))):
            flush()
            block = []
            continue
        m = re.match(r"^\s+([a-z_0-9]+ .*?)\s+//", line)
        if m:
            ins = m.group(1).strip()
            block.append(ins)
            if ins.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                flush()
                block = []
    flush()
    return hazards


def main():
    args = sys.argv[1:]
    strict = []
    while args and args[0] == "--strict":
        strict.append(args[1])
        args = args[2:]
    bad = 0
    for obj in args:
        with tempfile.TemporaryDirectory() as tmp:
            co = device_code(obj, tmp)
            if co is None:
                continue
            hz = audit(co)
        for k, lst in sorted(hz.items()):
            copies = [x for x in lst if x.startswith(("v_accvgpr", "v_mov"))]
            is_strict = any(s in k for s in strict)
            print("%s %s: %d vector instruction(s) in front of an EXEC restore (%d register copies)%s"
                  % (os.path.basename(obj), k[:90], len(lst), len(copies), "  <-- STRICT" if is_strict and copies else ""))
            for x in lst[:4]:
                print("      ", x)
            if is_strict and copies:
                bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

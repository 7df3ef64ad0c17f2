#!/usr/bin/env python3
"""Build-time repair + gate for a code-generation defect of the gfx950 back end (ROCm 7.2 / LLVM 22): register copies placed
at the TOP of a join block, IN FRONT of the instruction that restores EXEC.

Structured control flow is lowered to EXEC masking: a region is entered through `s_and_saveexec_b64 sN, cond` +
`s_cbranch_execz JOIN`, a divergent loop left through `s_andn2_b64 exec, exec, sN` + `s_cbranch_execnz LOOP`, and the join block
starts with `s_or_b64 exec, exec, sN`.  On both edges the join block is entered with EXEC == 0 for the lanes that skipped the
region.  The register allocator's live-range split copies (`v_mov`, `v_accvgpr_read/write`) and sunk instructions belong to the
join block -- every lane that reaches it must execute them -- but are sometimes emitted in front of that `s_or_b64`: the lanes
that skipped the region then never execute the copy and continue with whatever the destination register held before (an OLDER
VALUE OF THE SAME PROGRAM, not pre-kernel garbage: a register-poison test cannot see it).  Found in round 4 in
adj_kernel<LvUde<NetTanh32, 8>, Tsit5, fast>: the copy of `tstop` into its new register in front of the join of locate()'s
guarded loop was skipped by the lanes that did not enter the loop, the next step ran past a save time and the solve diverged
(tests/test_gpu_fuzz.py::test_random_corner_matches_oracle[3]; DESIGN.md "stale-register finding").

What this tool does to a device assembly file (`hipcc -S --cuda-device-only`):
  * finds every point that is entered with EXEC == 0 -- a label targeted by `s_cbranch_execz`, the fall-through of
    `s_cbranch_execnz` -- and the first EXEC-restoring instruction R (`s_or_b64 exec, exec, X` / `s_or_saveexec_b64 A, B`) of that
    straight-line block;
  * every EXEC-dependent vector instruction between the entry and R is moved BEHIND R (order kept), followed by `s_nop 4` (the
    hazard distances of the block were computed for the old order); scalar instructions and lane-indexed moves
    (v_readlane / v_writelane / v_readfirstlane: EXEC-independent) stay where they are;
  * the move is only done when it is provably order-independent: no register is shared between a moved and a not-moved
    instruction of the prefix (all operands of an instruction count as read AND written; implicit VCC / SCC included), R's mask
    register is not touched by a moved instruction, no memory instruction is moved across an s_waitcnt.  Anything else is reported
    as UNHANDLED and fails the build (exit code 2): a human has to look at it.

usage:  isa_endcf_fix.py in.s out.s        (repairs, prints a report, exit 2 on unhandled sites)
        isa_endcf_fix.py --audit file.s    (no output file: exit 1 if any site -- handled or not -- is present)"""
import re
import sys

VEC_PREFIX = ("v_", "ds_", "global_", "buffer_", "scratch_", "flat_", "tbuffer_")
LANE_OPS = ("v_readlane", "v_writelane", "v_readfirstlane", "v_nop")
MEM_PREFIX = ("ds_", "global_", "buffer_", "scratch_", "flat_", "tbuffer_")
TERMINATORS = ("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_trap")

RE_LABEL = re.compile(r"^(\.LBB\d+_\d+):")
RE_FUNC = re.compile(r"^([A-Za-z_][\w$.]*):")
RE_REG = re.compile(r"\b([vsa])\[(\d+):(\d+)\]|\b([vsa])(\d+)\b|\b(vcc|exec|m0|scc)(?:_lo|_hi)?\b")


def instr_of(line):
    """(opcode, operand text) of an instruction line, or None for labels / directives / comments / blank lines"""
    s = line.split(";")[0].strip() if not line.lstrip().startswith(";") else ""
    if not s or s.startswith(".") or s.endswith(":"):
        return None
    parts = s.split(None, 1)
    return parts[0], (parts[1] if len(parts) > 1 else "")


def regs_of(op, operands):
    out = set()
    for m in RE_REG.finditer(operands):
        if m.group(1):
            out.update("%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        elif m.group(4):
            out.add("%s%s" % (m.group(4), m.group(5)))
        else:
            out.add(m.group(6))
    if op.startswith("s_") and not op.startswith(("s_nop", "s_waitcnt", "s_mov", "s_load", "s_sleep")):
        out.add("scc")
    if op.startswith(("v_div_fmas", "v_cndmask_b32_e32", "v_addc", "v_subb", "v_subbrev")) or op.endswith("_e32") and op.startswith("v_cmp"):
        out.add("vcc")
    return out


def is_vector(op):
    return op.startswith(VEC_PREFIX) and not op.startswith(LANE_OPS)


def writes_exec(op, operands):
    if op.startswith("v_cmpx"):
        return True
    if not op.startswith("s_"):
        return False
    first = operands.split(",")[0].strip()
    return first.startswith("exec") or "saveexec" in op


def process(text, repair=True):
    lines = text.split("\n")
    n = len(lines)
    ins = [instr_of(l) for l in lines]
    # entries that are reached with EXEC == 0
    targets = set()
    for i, it in enumerate(ins):
        if it and it[0] == "s_cbranch_execz":
            targets.add(it[1].strip())
    entries = []
    for i, l in enumerate(lines):
        m = RE_LABEL.match(l.strip())
        if m and m.group(1) in targets:
            entries.append(i + 1)
        it = ins[i]
        if it and it[0] == "s_cbranch_execnz":
            entries.append(i + 1)
    report = {"fixed": [], "unhandled": []}
    func = [None] * n
    cur = None
    for i, l in enumerate(lines):
        m = RE_FUNC.match(l)
        if m and not m.group(1).startswith(".L"):
            cur = m.group(1)
        func[i] = cur
    done = set()
    edits = []  # (first line, R line, [vector line indices])
    for start in sorted(set(entries)):
        i = start
        own_label_ok = True
        prefix = []  # instruction line indices between the entry and R
        R = None
        bad = None
        while i < n:
            s = lines[i].strip()
            if RE_LABEL.match(s):
                if i == start or (not prefix and own_label_ok):  # the entry's own label (fall-through entry onto a labelled block)
                    i += 1
                    continue
                break  # another block starts: no restore in this straight-line piece
            it = ins[i]
            if it is None:
                i += 1
                continue
            own_label_ok = False
            op, operands = it
            if op.startswith(TERMINATORS):
                break
            if writes_exec(op, operands):
                saved = any(ins[j][0] == "s_mov_b64" and re.match(r"s\[\d+:\d+\]\s*,\s*exec\s*$", ins[j][1]) for j in prefix)
                if (op == "s_or_b64" and re.match(r"exec\s*,\s*exec\s*,", operands)) or op == "s_or_saveexec_b64":
                    R = i      # the join: EXEC restored from the mask saved at the region's entry
                elif op == "s_mov_b64" and not saved:
                    R = i      # (`s_mov_b64 exec, sN`: the same restore where EXEC is known to be 0)
                elif op in ("s_and_saveexec_b64", "s_andn2_b64", "s_and_b64") or (op == "s_mov_b64" and saved):
                    # a region ENTRY (EXEC narrowed further, possibly open-coded: s_mov sY, exec; s_and sX, sY, c; s_mov exec, sX):
                    # the lanes that arrive here with EXEC == 0 stay off until an outer join -- and what is defined in
                    # between belongs to variables that live inside the outer region, where those lanes are off anyway
                    # (the wave-level skip of a kernel body whose wavefront has no work)
                    pass
                else:
                    bad = "first EXEC write is `%s %s`" % (op, operands)
                break
            prefix.append(i)
            i += 1
        vec = [j for j in prefix if is_vector(ins[j][0])]
        if not vec:
            continue
        key = (vec[0], R)
        if key in done:
            continue
        done.add(key)
        where = "%s line %d" % (func[start], vec[0] + 1)
        if R is None:
            if bad:
                report["unhandled"].append((where, bad, [lines[j].strip() for j in vec[:6]]))
            continue  # (no EXEC restore in this block at all: the prefix belongs to lanes that are active -- not a join)
        # legality of moving `vec` behind R
        stay = [j for j in prefix if j not in vec and j > vec[0]]
        rregs = regs_of(*ins[R]) - {"exec", "scc"}
        why = None
        for j in vec:
            rj = regs_of(*ins[j])
            if rj & rregs:
                why = "`%s` touches the mask register of the restore" % lines[j].strip()
            if ins[j][0].startswith(MEM_PREFIX) and any(ins[k][0] == "s_waitcnt" for k in prefix if k > j):
                why = "memory instruction `%s` in front of an s_waitcnt" % lines[j].strip()
            for k in stay:
                if k > j and (rj - {"exec"}) & (regs_of(*ins[k]) - {"exec"}):
                    why = "`%s` and `%s` share a register" % (lines[j].strip(), lines[k].strip())
            if why:
                break
        if why:
            report["unhandled"].append((where, why, [lines[j].strip() for j in vec[:6]]))
            continue
        report["fixed"].append((where, lines[R].strip(), [lines[j].strip() for j in vec]))
        edits.append((vec, R))
    if repair and edits:
        moved = set()
        insert_after = {}
        for vec, R in edits:
            vec = [j for j in vec if j not in moved]
            moved.update(vec)
            insert_after.setdefault(R, []).extend(vec)
        out = []
        for i, l in enumerate(lines):
            if i in moved:
                continue
            out.append(l)
            if i in insert_after:
                out.extend(lines[j] for j in insert_after[i])
                out.append("\ts_nop 4")
        text = "\n".join(out)
    return text, report


def main():
    args = sys.argv[1:]
    if args and args[0] == "--audit":
        rc = 0
        for f in args[1:]:
            _, rep = process(open(f).read(), repair=False)
            for where, r, v in rep["fixed"]:
                print("%s: %s: %d vector instruction(s) in front of `%s`: %s" % (f, where, len(v), r, "; ".join(v[:4])))
                rc = 1
            for where, why, v in rep["unhandled"]:
                print("%s: %s: UNHANDLED (%s): %s" % (f, where, why, "; ".join(v[:4])))
                rc = 1
        return rc
    src, dst = args
    text, rep = process(open(src).read())
    open(dst, "w").write(text)
    for where, r, v in rep["fixed"]:
        print("endcf-fix: %s: moved %d instruction(s) behind `%s`: %s" % (where, len(v), r, "; ".join(v[:4])))
    for where, why, v in rep["unhandled"]:
        print("endcf-fix: UNHANDLED %s (%s): %s" % (where, why, "; ".join(v[:4])))
    # the repaired text must be clean
    _, again = process(text, repair=False)
    if again["fixed"] or rep["unhandled"]:
        return 2
    return 0


if __name__ == "__main__":
    sys.exit(main())

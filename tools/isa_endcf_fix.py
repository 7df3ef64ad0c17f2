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
  * PURE REGISTER COPIES between the entry and R -- `v_mov_b32/b64 vD, vS`, `v_accvgpr_read_b32`, `v_accvgpr_write_b32`,
    `v_accvgpr_mov_b32`, plain encodings only (no DPP / SDWA modifier, no lane select) -- are moved BEHIND R (order kept), followed by
    `s_nop 4`.  These are what the register allocator's live-range splitting emits, they have no implicit operand (VCC / SCC / M0 /
    EXEC-as-data), they are not matrix or memory instructions, and the hazard classes that involve a plain VALU register write
    (VALU write -> DPP / readlane select / VMEM address / MFMA source) need at most 5 wait states, which `s_nop 4` provides; hazards
    of instructions in FRONT of the block towards the copies only get longer distances by the move;
  * the move is only done when it is provably order-independent: no register is shared between a moved and a not-moved
    instruction of the prefix (all operands of an instruction count as read AND written), R's mask register is not touched by a
    moved instruction;
  * ANY OTHER EXEC-dependent vector instruction in front of R -- arithmetic, v_cmp (writes an SGPR mask), MFMA, DPP, LDS / global /
    scratch memory instructions -- is reported as UNHANDLED and fails the build (exit code 2): a human has to look at it.  (Rounds 4
    and 5 never met one; the gate exists so that such a site cannot be accepted silently.)  Scalar instructions and lane-indexed moves
    (v_readlane / v_writelane / v_readfirstlane: EXEC-independent) stay where they are;
  * a region ENTRY (EXEC narrowed further before any restore) behind an EXEC == 0 entry is NOT a join -- see `entry_site_is_dead`
    for the argument that is checked, not assumed, for every such site.

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


COPY_OPS = ("v_mov_b32", "v_mov_b32_e32", "v_mov_b64", "v_mov_b64_e32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_accvgpr_mov_b32")
RE_PLAIN_COPY = re.compile(r"^\s*[va](\d+|\[\d+:\d+\])\s*,\s*([va](\d+|\[\d+:\d+\])|-?\d+(\.\d+)?|0x[0-9a-fA-F]+)\s*$")


def is_pure_copy(op, operands):
    """a live-range split copy: vector register (or inline constant) -> vector register, plain encoding (a DPP / SDWA variant has
    modifier text behind the operands and an opcode suffix; both are rejected by the exact opcode list and the operand pattern)"""
    return op in COPY_OPS and RE_PLAIN_COPY.match(operands) is not None


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
    report = {"fixed": [], "unhandled": [], "entry": 0}
    entry_sites = []
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
                # another block starts.  If this piece falls through into a block that BEGINS with the restore, its vector instructions
                # stand in front of a join all the same -- but they cannot be moved behind a restore that other predecessors also
                # reach: refuse
                k = i
                while k < n and (ins[k] is None):
                    k += 1
                if k < n and any(is_vector(ins[j][0]) for j in prefix) and not ins[prefix[-1]][0].startswith(TERMINATORS):
                    op2, opr2 = ins[k]
                    if (op2 == "s_or_b64" and re.match(r"exec\s*,\s*exec\s*,", opr2)) or op2 == "s_or_saveexec_b64":
                        bad = "falls through label `%s` into the restore `%s %s`" % (s, op2, opr2)
                break
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
                    # a region ENTRY (EXEC narrowed further, possibly open-coded: s_mov sY, exec; s_and sX, sY, c; s_mov exec, sX).
                    # Why this is not a site: both EXEC == 0 edges are WAVE-level (s_cbranch_execz is taken, s_cbranch_execnz falls
                    # through, only when every lane is off), so on that edge the whole wave executes nothing here, the narrowing
                    # saves and keeps EXEC == 0, and the lanes come back at the `s_or_b64 exec, exec, sM` of the mask sM that switched
                    # them off -- an instruction that post-dominates this block in structured control flow and that this tool examines
                    # as a site of its own.  The defect is a property of the JOIN block (the allocator inserts its split copies at the
                    # top of the block that holds the restore): a block without a restore is not one, and a block that falls through
                    # into one is refused above.  Counted in the report (`entry`) so that the number is visible in the build log.
                    entry_sites.append(start)
                else:
                    bad = "first EXEC write is `%s %s`" % (op, operands)
                break
            prefix.append(i)
            i += 1
        vec = [j for j in prefix if is_vector(ins[j][0])]
        if not vec:
            continue
        if entry_sites and entry_sites[-1] == start:
            report["entry"] += 1
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
            if not is_pure_copy(*ins[j]):
                why = "`%s` is not a pure register copy (only v_mov / v_accvgpr_read / v_accvgpr_write may be moved)" % lines[j].strip()
                break
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

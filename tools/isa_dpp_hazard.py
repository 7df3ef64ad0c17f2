#!/usr/bin/env python3
"""Build-time pass for the hand-written DPP instructions of the Fisher-KPP vector kernel (csrc/ude_model_kpp_vec.h).

`v_fmac_f64_dpp ... row_newbcast:k` is written as inline assembly (the compiler has no way to select it: it emits a separate
`v_mov_b64_dpp` in front of a plain v_fma_f64).  The back end's hazard recogniser does not look inside an asm statement, so two
software-managed hazards of gfx9-class hardware are not padded there (LLVM's GCNHazardRecognizer::checkDPPHazards does it for the
DPP instructions the compiler itself selects):

  * a VALU instruction writes a VGPR, a DPP instruction reads THAT VGPR as its DPP operand (src0): 2 wait states in between.
    The DPP operands here are the network's weight registers, written once when a kernel starts -- until the register allocator
    parks one in an AGPR and brings it back with `v_accvgpr_read_b32` directly in front of the asm statement (seen the first time
    the kernel was built with `-amdgpu-mfma-vgpr-form`: two sites, wrong gradients).
  * a VALU instruction writes EXEC (v_cmpx*), a DPP instruction follows: 5 wait states.

What the pass does to a device assembly text: for every `*_dpp` instruction INSIDE an asm statement (`;;#ASMSTART` .. `;;#ASMEND`)
it walks back over the preceding instructions of the straight-line code, counts wait states (every instruction is one, `s_nop N` is
N + 1) and, if a conflicting write is closer than required -- or a label lies inside the window (another predecessor could end in
anything) --, inserts the missing `s_nop` in front of the asm statement.  It refuses (exit code 2) a DPP control other than
`row_newbcast` on a 64-bit instruction (the assembler refuses it too) and an instruction it cannot parse.

usage: isa_dpp_hazard.py in.s out.s      isa_dpp_hazard.py --audit file.s   (exit 1 if a hazard is present)"""
import re
import sys

RE_LABEL = re.compile(r"^(\.LBB\d+_\d+|[A-Za-z_][\w$.]*):")
RE_VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
VGPR_WAIT = 2   # VALU write of the DPP operand -> DPP read
EXEC_WAIT = 5   # VALU write of EXEC -> DPP


def vregs(tok):
    out = set()
    for m in RE_VREG.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def instr_of(line):
    s = line.split(";")[0].strip() if not line.lstrip().startswith(";") else ""
    if not s or s.startswith(".") or s.endswith(":"):
        return None
    parts = s.split(None, 1)
    return parts[0], (parts[1] if len(parts) > 1 else "")


def process(text, repair=True):
    lines = text.split("\n")
    report = {"inserted": [], "errors": [], "dpp": 0}
    in_asm = False
    inserts = {}   # line index of the asm statement's ;;#ASMSTART -> nops needed
    asm_start = None
    for i, l in enumerate(lines):
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm, asm_start = True, i
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        it = instr_of(l)
        if not in_asm or it is None or not it[0].endswith("_dpp"):
            continue
        op, operands = it
        report["dpp"] += 1
        toks = [t.strip() for t in operands.split(",")]
        if len(toks) < 2:
            report["errors"].append((i + 1, "cannot parse `%s`" % s))
            continue
        if "f64" in op or "b64" in op:
            if "row_newbcast:" not in operands:
                report["errors"].append((i + 1, "64-bit DPP with a control other than row_newbcast: `%s`" % s))
                continue
        src0 = vregs(toks[1].split()[0])
        # walk back: wait states between a conflicting writer and this instruction
        need = 0
        ws = 0
        j = i - 1
        while j >= 0 and ws < EXEC_WAIT:
            t = lines[j].strip()
            if RE_LABEL.match(t):
                # a block boundary inside the window: the other predecessors are unknown -- pad for the register hazard
                if ws < VGPR_WAIT:
                    need = max(need, VGPR_WAIT - ws)
                break
            pj = instr_of(lines[j])
            if pj is None:
                j -= 1
                continue
            pop, popr = pj
            if pop == "s_nop":
                ws += int(popr.strip() or "0", 0) + 1
                j -= 1
                continue
            if pop.startswith("v_cmpx") and ws < EXEC_WAIT:
                need = max(need, EXEC_WAIT - ws)
            if pop.startswith("v_") and ws < VGPR_WAIT:
                dst = popr.split(",")[0]
                if vregs(dst) & src0:
                    need = max(need, VGPR_WAIT - ws)
            ws += 1
            j -= 1
        if need:
            at = asm_start if asm_start is not None else i
            inserts[at] = max(inserts.get(at, 0), need)
            report["inserted"].append((i + 1, need, s))
    if repair and inserts:
        out = []
        for i, l in enumerate(lines):
            if i in inserts:
                out.append("\ts_nop %d" % (inserts[i] - 1))
            out.append(l)
        text = "\n".join(out)
    return text, report


def main():
    args = sys.argv[1:]
    if args and args[0] == "--audit":
        rc = 0
        for f in args[1:]:
            _, rep = process(open(f).read(), repair=False)
            for ln, need, s in rep["inserted"]:
                print("%s:%d: DPP hazard, %d wait state(s) missing: %s" % (f, ln, need, s))
                rc = 1
            for ln, why in rep["errors"]:
                print("%s:%d: %s" % (f, ln, why))
                rc = 1
        return rc
    src, dst = args
    text, rep = process(open(src).read())
    open(dst, "w").write(text)
    for ln, need, s in rep["inserted"]:
        print("dpp-hazard: line %d: s_nop %d in front of `%s`" % (ln, need - 1, s))
    for ln, why in rep["errors"]:
        print("dpp-hazard: ERROR line %d: %s" % (ln, why))
    _, again = process(text, repair=False)
    return 2 if rep["errors"] or again["inserted"] else 0


if __name__ == "__main__":
    sys.exit(main())

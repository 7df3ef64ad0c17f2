"""Instruction mix per kernel of a gfx950 assembly file (hipcc -S / -save-temps): how many issue slots of which kind.
usage: python tools/isa_mix.py file.s [substring-of-kernel-name ...]"""
import re
import sys
from collections import Counter


def kernels(txt):
    out = []
    cur = None
    for line in txt.split("\n"):
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            cur = [m.group(1), []]
            out.append(cur)
        elif cur is not None and line.startswith("\t") and not line.strip().startswith((".", ";")):
            op = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", line.strip().split()[0])
            cur[1].append(op)
            if op == "s_endpgm":
                cur = None
    return out


def mix(ops):
    c = Counter(ops)
    g = lambda pred: sum(v for k, v in c.items() if pred(k))
    return dict(total=len(ops), valu=g(lambda k: k.startswith("v_") and not k.startswith("v_mfma")), mfma=g(lambda k: k.startswith("v_mfma")),
                fma64=c["v_fma_f64"] + c["v_fmac_f64"], smem=g(lambda k: k.startswith("s_load") or k.startswith("s_buffer_load")), lds=g(lambda k: k.startswith("ds_")),
                vmem=g(lambda k: k.startswith(("global_", "buffer_", "flat_", "scratch_"))), salu=g(lambda k: k.startswith("s_") and not k.startswith(("s_load", "s_buffer_load", "s_waitcnt", "s_nop"))),
                waitcnt=c["s_waitcnt"], nop=c["s_nop"], lane_rw=c["v_readlane_b32"] + c["v_writelane_b32"] + c["v_readfirstlane_b32"],
                acc_mov=c["v_accvgpr_read_b32"] + c["v_accvgpr_write_b32"] + c["v_accvgpr_mov_b32"], vmov=c["v_mov_b32"] + c["v_mov_b64"], cndmask=c["v_cndmask_b32"], rcp=c["v_rcp_f64"])


if __name__ == "__main__":
    txt = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    for name, ops in kernels(txt):
        if len(ops) < 50 or (pats and not any(p in name for p in pats)):
            continue
        print(name[:150])
        print("   ", mix(ops))

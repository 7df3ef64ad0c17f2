#!/usr/bin/env python3
"""Static look at a kernel's ISA (hipcc -S --cuda-device-only output): size, loops (by back edges) and instruction mix.
usage: python tools/isa_loops.py file.s kernel_substring [min_loop_instrs]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
want = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 100
start = [i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l) and want in l]
PRE = ["scratch_load", "scratch_store", "v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64", "v_accvgpr", "v_cndmask", "v_mov",
       "ds_bpermute", "ds_read", "ds_write", "s_waitcnt", "s_barrier", "v_readlane", "v_div", "v_rcp", "global_load",
       "global_store", "s_load", "s_nop", "s_cbranch", "v_cmp"]


def mix(lines):
    c = collections.Counter()
    for l in lines:
        m = re.match(r"^\s*([a-z_0-9]+)", l)
        if m and not l.strip().startswith((".", ";")):
            k = m.group(1)
            for p in PRE:
                if k.startswith(p):
                    k = p
                    break
            c[k] += 1
    return c


for s in start:
    e = next(i for i in range(s, len(txt)) if txt[i].strip().startswith(".size"))
    lines = txt[s:e]
    c = mix(lines)
    print("kernel %s...: %d instructions (~%d KB)" % (txt[s][:60], sum(c.values()), sum(c.values()) * 7 // 1024))
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = set()
    for i, l in enumerate(lines):
        m = re.match(r"^\s*(s_cbranch\w*|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(2), 1 << 30) < i:
            loops.add((labels[m.group(2)], i))
    for a, b in sorted(loops):
        c = mix(lines[a:b])
        n = sum(c.values())
        if n >= minlen:
            print("  loop lines %d..%d: %d instrs: %s" % (a, b, n, ", ".join("%s %d" % kv for kv in c.most_common(18))))

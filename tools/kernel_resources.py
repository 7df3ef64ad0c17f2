#!/usr/bin/env python3
"""Register / scratch / occupancy of the shipped kernels, read from the build logs.

`build.py` compiles every translation unit with `-Rpass-analysis=kernel-resource-usage` and keeps the remarks in
`universal_differential_equations_amd/build/<unit>.log`.  This tool turns them into the table DESIGN.md cites
(`profiles/r06_kernel_resources.md`): the numbers in the documentation are the numbers of the objects that were linked, not typed.

    tools/kernel_resources.py <log>                 one log, all kernels (developer use)
    tools/kernel_resources.py --table [--write]     the table of the dominant kernels of every workload (stdout, or the committed file)

tests/test_kernel_resources_cpu.py regenerates the table where a build/ directory exists and compares it with the committed file."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "universal_differential_equations_amd", "build")
TABLE = os.path.join(ROOT, "profiles", "r06_kernel_resources.md")
KEYS = ("VGPRs:", "AGPRs:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:", "SGPRs:", "LDS Size [bytes/block]:", "VGPRs Spill:")

# (workload, build log, glob of the demangled kernel name): the kernels bench.py's lines and DESIGN.md talk about
ROWS = [
    ("lv (headline) forward", "ude_inst_lv_s1n_g5_w1_tsit5.log", "fwd_kernel<*false, double, false>"),
    ("lv (headline) adjoint", "ude_inst_lv_s1n_g5_w1_tsit5.log", "adj_kernel<*false, 1, double>"),
    ("lv discrete sweep", "ude_inst_lv_s1n_g5_w1_tsit5.log", "dadj_kernel<*false, double>"),
    ("lv_tanh32 adjoint", "ude_inst_lv_tanh32_g16_w1_tsit5.log", "adj_kernel<*false, 1, double>"),
    ("lv_shape8 adjoint (run-time shapes on 8-lane groups)", "ude_inst_lv_rt4_g8_w1_tsit5.log", "adj_kernel<*false, 1, double>"),
    ("lv_wave64 adjoint (runtime shapes)", "ude_inst_generic_2_l4_g64_w1_tsit5.log", "adj_kernel<*false, 1, double>"),
    ("seir forward (lock-step)", "ude_seir_ls.log", "seir_ls_fwd_kernel<Vern7Tab, false>*"),
    ("seir adjoint, parity mode (lock-step, second generation)", "ude_seir_ls.log", "seir_ls2_adj_kernel<Vern7Tab, false>*"),
    ("runtime-shape exposure chain 3-H1-H2-1, forward (lock-step, GEN)", "ude_seir_ls.log", "seir_ls_fwd_kernel<Vern7Tab, true>*"),
    ("runtime-shape exposure chain 3-H1-H2-1, adjoint (lock-step, GEN)", "ude_seir_ls.log", "seir_ls2_adj_kernel<Vern7Tab, true>*"),
    ("seir adjoint, fast mode (lock-step, block-level MFMA accumulation)", "ude_seir_lsf.log", "seir_lsf_adj_kernel<Vern7Tab, false>*"),
    ("runtime-shape exposure chain 3-H1-H2-1, adjoint, fast mode (lock-step, GEN)", "ude_seir_lsf.log", "seir_lsf_adj_kernel<Vern7Tab, true>*"),
    ("node forward (lock-step)", "ude_node_ls.log", "node_ls_fwd_kernel<Vern7Tab>*"),
    ("node adjoint, parity mode (lock-step, second generation)", "ude_node_ls.log", "node_ls2_adj_kernel<Vern7Tab>*"),
    ("node adjoint, fast mode (lock-step, block-level MFMA accumulation)", "ude_node_lsf.log", "node_lsf_adj_kernel<Vern7Tab>*"),
    ("kpp forward (1024 points; round 6: network on the vector unit, DPP-broadcast weights)", "ude_inst_kpp_ude_1024_g256_w1_tsit5.log", "fwd_kernel<*false, double, false>"),
    ("kpp adjoint (1024 points; round 6: vector network + packed matrix-core contraction)", "ude_inst_kpp_ude_1024_g256_w1_tsit5.log", "adj_kernel<*false, 1, double>"),
    ("kpp adjoint (1024 points), Vern7 (half-size transposition tile)", "ude_inst_kpp_ude_1024_g256_w1_vern7.log", "adj_kernel<*false, 1, double>"),
    ("kpp adjoint (1024 points), run-time-shape reaction network", "ude_inst_kpp_rt_1024_g256_w1_tsit5.log", "adj_kernel<*false, 1, double>"),
    ("hjb forward", "ude_hjb.log", "hjb_fwd_kernel*"),
    ("hjb backward", "ude_hjb.log", "hjb_bwd_kernel*"),
]


def demangle(name):
    try:
        out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        out = name
    out = out.replace("ude::", "").replace("void ", "")
    out = re.sub(r"NetCfg<IntList<([\d, ]+)>, IntList<([\d, ]+)>\s*>",
                 lambda m: "Net[%s|%s]" % (m.group(1).replace(" ", ""), m.group(2).replace(" ", "")), out)
    return re.sub(r"\(KParams\)", "", out)


def parse(path):
    """[{name, VGPRs:, AGPRs:, ...}] of one build log (the first occurrence of a kernel counts: a log may hold a unit twice)"""
    rows, cur, seen = [], None, set()
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"mangled": m.group(1)}
            if m.group(1) in seen:
                cur = None
                continue
            seen.add(m.group(1))
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key in KEYS:
            if key in line:
                cur[key] = line.split(key)[1].split("[")[0].strip()
    for r in rows:
        r["name"] = demangle(r["mangled"])
    return rows


def table():
    out = ["# Kernel resources of the shipped objects (generated: tools/kernel_resources.py --table --write)", "",
           "Read from `universal_differential_equations_amd/build/<unit>.log` (`-Rpass-analysis=kernel-resource-usage` of the very compile that",
           "produced the linked object).  Registers are per lane; occupancy = wavefronts per SIMD the register file allows.", "",
           "| kernel | build log | VGPRs | AGPRs | scratch B/lane | VGPR spills | occupancy |", "|---|---|---|---|---|---|---|"]
    for what, log, sub in ROWS:
        path = os.path.join(BUILD, log)
        if not os.path.exists(path):
            out.append("| %s | %s | (not built) | | | | |" % (what, log))
            continue
        import fnmatch
        hit = [r for r in parse(path) if fnmatch.fnmatchcase(r["name"], "*" + sub)]
        if not hit:
            out.append("| %s | %s | (no kernel matching `%s`) | | | | |" % (what, log, sub))
            continue
        r = hit[0]
        out.append("| %s | %s | %s | %s | %s | %s | %s |" % (what, log, r.get("VGPRs:"), r.get("AGPRs:"), r.get("ScratchSize [bytes/lane]:"),
                                                           r.get("VGPRs Spill:"), r.get("Occupancy [waves/SIMD]:")))
    return "\n".join(out) + "\n"


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--table":
        txt = table()
        if "--write" in sys.argv:
            open(TABLE, "w").write(txt)
        else:
            sys.stdout.write(txt)
        return 0
    for r in parse(sys.argv[1]):
        print(r["name"][:88].ljust(88), "VGPR", r.get("VGPRs:"), "AGPR", r.get("AGPRs:"), "SGPR", r.get("SGPRs:"), "scratch",
              r.get("ScratchSize [bytes/lane]:"), "occ", r.get("Occupancy [waves/SIMD]:"))
    return 0


if __name__ == "__main__":
    sys.exit(main())

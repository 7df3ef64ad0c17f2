"""Build libudecore.so (HIP, gfx950) in-tree.  `python -m universal_differential_equations_amd.build`

Every (model, lanes-per-trajectory, algorithm) kernel instance is its own translation unit
(csrc/ude_inst.hip compiled with -D flags) so hipcc runs them in parallel; objects are rebuilt only
when a source is newer.  The instance list below is mirrored into csrc/ude_instances_gen.h, which
udecore.hip includes to build its dispatch table.
"""
import hashlib
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libudecore.so")
# debug build of the two host-side translation units (-DUDE_DEBUG_HOOKS: register/LDS poison kernels, workspace fill,
# deep-BSDE phase clocks) linked with the SAME kernel objects: what tests/test_gpu_poison.py loads.  Nothing of it is
# in libudecore.so.
LIB_DBG = os.path.join(HERE, "libudecore_dbg.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: ARITH-SPEC -- every fma in the kernels is written explicitly, nothing else may fuse
# -fhip-fp32-correctly-rounded-divide-sqrt (the default, stated): f32 `/` and sqrtf are IEEE -- the Float32 path relies on it
# (HIP's __fsqrt_rn intrinsic is the NATIVE approximate sqrt unless OCML_BASIC_ROUNDED_OPERATIONS is defined: not used)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"]

# (model id, C++ model type, lanes per trajectory, kernel variant VAR [, block threads [, algorithms]]).  VAR is a template
# parameter of adj_kernel (1 = default; 2 = __launch_bounds__ for two waves per SIMD; 9 = lambda-only timing experiment):
# variants must differ in their template arguments, a macro-only difference gives one mangled name and the linker keeps one body
MODELS = [
    ("MID_LV_TRUE", "LvTrue<1>", 1, 1),
    ("MID_LV_S1", "LvUde<NetS1,1>", 1, 1),
    ("MID_LV_S1", "LvUde<NetS1,5>", 5, 1),
    ("MID_LV_S1N", "LvUde<NetS1,5,0>", 5, 1),
    # (6 lanes -- 10 trajectories per wavefront, 1000 wavefronts, still one round for the 10k ensemble -- was measured in round 3:
    #  adj_kernel 1.46 ms against 1.38 ms with 5 lanes; not kept)
    ("MID_LV_S1", "LvUde<NetS1,4>", 4, 1),
    ("MID_LV_S1", "LvUde<NetS1,8>", 8, 1),
    ("MID_LV_HUDSON", "LvUde<NetHudson,8>", 8, 1),
    # run-time shapes of the LV kind (two / three hidden layers of width <= 8, any activation): padded register copy of the weights
    ("MID_LV_RT2", "LvUde<NetLvRt2,8>", 8, 1),
    ("MID_LV_RT2_W16", "LvUde<NetLvRt2W16,16>", 16, 1),
    ("MID_LV_RT3", "LvUde<NetLvRt3,8>", 8, 1),
    ("MID_LV_RT4", "LvUde<NetLvRt4,8>", 8, 1),
    ("MID_LV_RT3_W5", "LvUde<NetLvRt3W5,5>", 5, 1),
    ("MID_LV_RT4_W5", "LvUde<NetLvRt4W5,5>", 5, 1),
    ("MID_LV_RT3_W16", "LvUde<NetLvRt3W16,16>", 16, 1),
    ("MID_LV_RT4_W16", "LvUde<NetLvRt4W16,16>", 16, 1),
    ("MID_LV_TANH32", "LvUde<NetTanh32,8>", 8, 1),
    ("MID_LV_TANH32", "LvUde<NetTanh32,16>", 16, 1),
    ("MID_LV_TANH32", "LvUde<NetTanh32,32>", 32, 1),
    ("MID_SEIR_TRUE", "SeirTrue<1>", 1, 1),
    ("MID_SEIR_UDE", "SeirUde<64>", 64, 1, 256),
    ("MID_SEIR_UDE", "SeirUde<256>", 256, 1, 256),
    ("MID_SEIR_NODE", "SeirNode<64>", 64, 1, 256),
    ("MID_KPP_TRUE_32", "KppTrue<32,1>", 32, 1, 32),
    ("MID_KPP_TRUE_1024", "KppTrue<64,16>", 64, 1),
    ("MID_KPP_UDE_32", "KppUde<NetKpp,32,1>", 32, 1, 32),
    ("MID_KPP_S3_32", "KppUde<NetKppS3,32,1>", 32, 1, 32),
    ("MID_KPP_SMALL_32", "KppUde<NetKppSmall,32,1>", 32, 1, 32),
    ("MID_KPP_SMALL1_32", "KppUde<NetKppSmall1,32,1>", 32, 1, 32),
    ("MID_KPP_SMALL2_32", "KppUde<NetKppSmall2,32,1>", 32, 1, 32),
    # runtime-shape fallback: one wavefront per trajectory, layer sizes and activations are kernel arguments
    ("MID_GENERIC_2", "GenericUde<2>", 64, 1, 64, None, ("-DUDE_INST_GENERIC=1",)),
    ("MID_GENERIC_7", "GenericUde<7>", 64, 1, 64, None, ("-DUDE_INST_GENERIC=1",)),
    ("MID_GENERIC_2_L4", "GenericUde<2,4>", 64, 1, 64, None, ("-DUDE_INST_GENERIC=1",)),
    ("MID_GENERIC_7_L4", "GenericUde<7,4>", 64, 1, 64, None, ("-DUDE_INST_GENERIC=1",)),
    # ... and for Float32 LV-kind problems (hudson_bay.jl:77-104 with any FastChain)
    ("MID_GENERIC_2_F32", "GenericUde<2>", 64, 1, 64, None, ("-DUDE_INST_GENERIC=1", "-DUDE_F32=1")),
    ("MID_GENERIC_2_L4_F32", "GenericUde<2,4>", 64, 1, 64, None, ("-DUDE_INST_GENERIC=1", "-DUDE_F32=1")),
    # runtime-shape pointwise reaction network (any chain 1 -> .. -> 1 of <= 4 layers, width <= 32) on grids of <= 32 points
    ("MID_KPP_GENERIC_32", "KppGenericUde<32,1>", 32, 1, 32, None, ("-DUDE_INST_KPPGEN=1",)),
    ("MID_KPP_UDE_1024", "KppUde<NetKpp,64,16>", 64, 1, 64, ("UDE_ALG_TSIT5",)),
    # round 6: the network on the vector unit (weights broadcast out of registers by DPP), packed matrix-core contraction
    # (csrc/ude_model_kpp_vec.h); Vern7's ten stage rows leave the LDS room for the half-size transposition tile only
    ("MID_KPP_UDE_1024", "KppUdeV<NetKpp>", 256, 1, 256, ("UDE_ALG_TSIT5",)),
    ("MID_KPP_UDE_1024", "KppUdeV<NetKpp,4,32>", 256, 1, 256, ("UDE_ALG_VERN7",)),
    # run-time shape of the reaction network on the large grids (three tanh layers of width <= 16): padded operand tables
    ("MID_KPP_RT_1024", "KppUdeW<NetKppRt16>", 256, 1, 256, ("UDE_ALG_TSIT5",)),
    # Float32 problems (-DUDE_F32: the same kernels with real = float)
    ("MID_LV_HUDSON_F32", "LvUde<NetHudson,8>", 8, 1, 64, None, ("-DUDE_F32=1",)),
    ("MID_LV_RT3_F32", "LvUde<NetLvRt3,8>", 8, 1, 64, None, ("-DUDE_F32=1",)),
    ("MID_LV_RT4_F32", "LvUde<NetLvRt4,8>", 8, 1, 64, None, ("-DUDE_F32=1",)),
    ("MID_LV_RT3_W5_F32", "LvUde<NetLvRt3W5,5>", 5, 1, 64, None, ("-DUDE_F32=1",)),
    ("MID_LV_RT4_W5_F32", "LvUde<NetLvRt4W5,5>", 5, 1, 64, None, ("-DUDE_F32=1",)),
    ("MID_KPP_TRUE_32_F32", "KppTrue<32,1>", 32, 1, 32, None, ("-DUDE_F32=1",)),
    ("MID_KPP_S3_32_F32", "KppUde<NetKppS3,32,1>", 32, 1, 32, None, ("-DUDE_F32=1",)),
]
TABS = [("UDE_ALG_TSIT5", "Tsit5Tab"), ("UDE_ALG_VERN7", "Vern7Tab")]


def instances():
    out = []
    for row in MODELS:
        mid, model, g, w = row[:4]
        block = row[4] if len(row) > 4 else 64
        algs = row[5] if len(row) > 5 else None
        extra = list(row[6]) if len(row) > 6 else []
        for alg, tab in TABS:
            if algs and alg not in algs:
                continue
            name = "ude_inst_%s_g%d_w%d_%s" % (mid[4:].lower(), g, w, tab[:-3].lower())
            out.append(dict(name=name, mid=mid, model=model, g=g, w=w, alg=alg, tab=tab, block=block, extra=extra))
    return out


def write_instances_header():
    lines = ["// GENERATED by build.py -- kernel instance table", "#pragma once"]
    for i in instances():
        lines.append('extern "C" void %s(ude::Launch*);' % i["name"])
    lines.append("#define UDE_INSTANCE_TABLE \\")
    for i in instances():
        lines.append("    {%s, %s, %d, %d, %s}, \\" % (i["mid"], i["alg"], i["g"], i["w"], i["name"]))
    lines.append("")
    path = os.path.join(CSRC, "ude_instances_gen.h")
    txt = "\n".join(lines) + "\n"
    if not os.path.exists(path) or open(path).read() != txt:
        open(path, "w").write(txt)


HEADER = os.path.join(os.path.dirname(HERE), "include", "udecore.h")
DBG_EXPORTS = ["ude_dbg_poison_selftest"]  # debug library only (tests/test_gpu_poison.py: the positive control of the poison hook)


def header_exports():
    """the entry points include/udecore.h declares"""
    import re
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"^(?:int|void|const char\s*\*)\s*(ude_\w+)\s*\(", txt, flags=re.M)))


def write_version_script(path, extra):
    txt = "{\n  global:\n" + "".join("    %s;\n" % n for n in header_exports() + list(extra)) + "  local: *;\n};\n"
    if not os.path.exists(path) or open(path).read() != txt:
        open(path, "w").write(txt)
    return path


def newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))


def newest_dep(depfile):
    """newest mtime among the in-tree files a translation unit really includes (compiler-written depfile, -MD); without a
    depfile: the newest header of csrc/"""
    try:
        txt = open(depfile).read().replace("\\\n", " ")
    except OSError:
        return newest_header()
    t = 0.0
    root = os.path.dirname(HERE)
    for tok in txt.split()[1:]:
        if tok.startswith(root) or not tok.startswith("/"):
            try:
                t = max(t, os.path.getmtime(tok))
            except OSError:
                return float("inf")  # a dependency disappeared: rebuild
    return t


def write_poison_header():
    """ude_poison_gen.h (6.5k lines of generated inline assembly for the debug build) is produced here, not committed"""
    import importlib.util
    path = os.path.join(OBJ, "ude_poison_gen.h")
    gen = os.path.join(os.path.dirname(HERE), "tools", "gen_poison_header.py")
    if os.path.exists(path) and os.path.getmtime(path) > os.path.getmtime(gen):
        return
    spec = importlib.util.spec_from_file_location("gen_poison_header", gen)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.write(path)


LLVM_BIN = os.environ.get("UDE_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ENDCF_FIX = os.path.join(os.path.dirname(HERE), "tools", "isa_endcf_fix.py")
DPP_HAZARD = os.path.join(os.path.dirname(HERE), "tools", "isa_dpp_hazard.py")


def _endcf():
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_endcf_fix", ENDCF_FIX)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _dpp():
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_dpp_hazard", DPP_HAZARD)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def compile_one(job):
    """One translation unit -> host object with the gfx950 code object embedded.  NOT a plain `hipcc -c`: the device side goes
    through its assembly text, where tools/isa_endcf_fix.py repairs a code-generation defect of this compiler drop (vector
    register copies emitted in front of the EXEC restore of a join block: the lanes that skipped the region keep a stale value
    -- DESIGN.md 2a) and REFUSES the build if a site is left that it cannot prove safe to repair:
        hipcc --cuda-device-only -S  ->  isa_endcf_fix  ->  clang -x assembler  ->  lld  ->  clang-offload-bundler
        hipcc --cuda-host-only -fcuda-include-gpubinary <bundle>
    (the same five steps `hipcc -c` runs internally, see `hipcc -###`)."""
    src, obj, defs, log = job
    dep = obj + ".d"
    base = [HIPCC] + FLAGS + defs
    # an object is current only if it is newer than the files it includes AND was built by this exact command line (an
    # edited MODELS row / flag keeps the instance name but must not keep the object) AND by this version of the repair tool
    stamp = obj + ".cmd"
    sig = hashlib.sha256((" ".join(base + [src]) + open(ENDCF_FIX).read() + open(DPP_HAZARD).read()).encode()).hexdigest()
    if (os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_dep(dep))
            and os.path.exists(stamp) and open(stamp).read() == sig):
        return obj, 0, "up to date"
    stem = obj[:-2]
    asm, fixed, dev_o, hsaco, fatbin = stem + ".s", stem + ".fixed.s", stem + ".dev.o", stem + ".hsaco", stem + ".hipfb"
    out = []
    # the object carries the time its compile STARTED (os.utime below): a header edited while a minutes-long unit was compiling is
    # then newer than the object, and the unit is rebuilt -- round 6 found second-allocation objects compiled from the headers of
    # five minutes earlier that the plain mtime comparison called current (kernels missing a template parameter: a launch aborted)
    t_start = time.time()

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        out.append(r.stderr)
        return r.returncode

    rc = run(base + ["--cuda-device-only", "-S", "-Rpass-analysis=kernel-resource-usage", src, "-o", asm])
    if rc == 0:
        text, rep = _endcf().process(open(asm).read())
        for where, r, v in rep["fixed"]:
            out.append("endcf-fix: %s: moved %d instruction(s) behind `%s`: %s\n" % (where, len(v), r, "; ".join(v[:6])))
        for where, why, v in rep["unhandled"]:
            out.append("endcf-fix: UNHANDLED %s (%s): %s\n" % (where, why, "; ".join(v[:6])))
        _, again = _endcf().process(text, repair=False)
        if rep["unhandled"] or again["fixed"] or again["unhandled"]:
            rc = 2
        # second pass: the hand-written DPP instructions (inline assembly: the compiler's hazard recogniser does not see them) get
        # the wait states their operands need (tools/isa_dpp_hazard.py; csrc/ude_model_kpp_vec.h is the only user)
        text, drep = _dpp().process(text)
        for ln, need, what in drep["inserted"]:
            out.append("dpp-hazard: line %d: s_nop %d in front of `%s`\n" % (ln, need - 1, what))
        for ln, why in drep["errors"]:
            out.append("dpp-hazard: ERROR line %d: %s\n" % (ln, why))
        if drep["errors"] or _dpp().process(text, repair=False)[1]["inserted"]:
            rc = 2
        open(fixed, "w").write(text)
    if rc == 0:
        rc = run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", fixed, "-o", dev_o])
    if rc == 0:
        rc = run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, dev_o])
    if rc == 0:
        rc = run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
                  "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + hsaco, "-output=" + fatbin])
    if rc == 0:
        # (the dependency file is written HERE: the driver ignores -MD / -MF on the `-S --cuda-device-only` step -- round 5 found the
        #  .d files of round 3 still in use, blind to every header added since; the host pass includes the same files)
        rc = run(base + ["--cuda-host-only", "-Wno-unused-command-line-argument", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin, "-MD", "-MF", dep,
                         "-c", src, "-o", obj])
    open(log, "w").write("".join(out))
    for tmp in (asm, dev_o, hsaco, fatbin):  # (the repaired assembly stays: tests/test_build_gate_cpu.py audits it)
        if os.path.exists(tmp):
            os.remove(tmp)
    if rc == 0:
        os.utime(obj, (t_start, t_start))
        open(stamp, "w").write(sig)
    return obj, rc, "".join(out) if rc else ""


# A SECOND, INDEPENDENT detector for the stale-value defect class of DESIGN.md 2a (a register that keeps an OLDER value of the same
# kernel because a copy was skipped): every translation unit once more, through the same pipeline and the same gate, with a
# DIFFERENT REGISTER ALLOCATOR (LLVM's "basic" allocator for the vector registers instead of "greedy": other live-range splitting,
# other copies, other registers).  A stale-value bug moves with the allocation; tests/test_gpu_ra2.py runs the oracle-comparing fuzz /
# parity files against libudecore_ra2.so -- two allocations that both agree with the oracle bit for bit on every corner is evidence
# the poison test cannot give.  Not shipped, not loaded by anything but that test (UDE_LIB_VARIANT=ra2).
OBJ_RA2 = os.path.join(HERE, "build", "ra2")
LIB_RA2 = os.path.join(HERE, "libudecore_ra2.so")
RA2_FLAGS = ["-mllvm", "-vgpr-regalloc=basic"]


# A THIRD detector, one that shares NOTHING with the assembly pipeline above (round-5 review, item 8: the second allocation is gated by
# the same tools/isa_endcf_fix.py, so a defect of the rewriter common to both builds is invisible to it).  One representative
# translation unit per kernel family is compiled by a PLAIN `hipcc -c -O0` -- no -S, no rewriter, no hazard pass, the compiler's own
# assembler, and NO OPTIMISATION: at -O0 LLVM allocates registers with its "fast" allocator, which has no live-range splitting (every
# value that lives across a block boundary is stored to scratch where it is defined and reloaded in front of its use), so the defect
# class of DESIGN.md 2a (a SPLIT COPY emitted in front of a join block's EXEC restore) cannot be generated at all -- the work-around
# is structural, not a repair -- and neither the machine scheduler nor any other -O3 pass has touched the code.  The objects replace
# their shipping counterparts in libudecore_nrw.so (every other object is the shipping one); tests/test_gpu_ra2.py runs the
# oracle-comparing files against it: bit-identity with the oracle per trajectory = bit-identity with the rewritten shipping objects.
# Kilobytes of scratch per lane: a checker, never shipped, never timed.
# (Measured first and NOT usable: `-O3 -mllvm -vgpr-regalloc=fast` -- the fast allocator behind the optimising pipeline.  It compiles,
#  finds no site, and computes wrong results: the LV kernel fails on scenario_2's wiring only, the deep-BSDE forward kernel everywhere,
#  while -O0 builds of the same units are bit-identical to the oracle -- a code-generation problem of that unsupported combination,
#  not of these sources; profiles/r06_probes.md.)
# (csrc/ude_model_kpp_vec.h is not represented: its hand-written DPP instructions NEED the hazard pass; the 32-point instance of the
#  same model family is.)
OBJ_NRW = os.path.join(HERE, "build", "nrw")
LIB_NRW = os.path.join(HERE, "libudecore_nrw.so")
NRW_FLAGS = ["-O0"]   # (after FLAGS' -O3: the last -O wins)
NRW_UNITS = ["udecore", "ude_hjb", "ude_seir_ls", "ude_node_ls", "ude_seir_lsf", "ude_node_lsf",
             "ude_inst_lv_s1_g5_w1_tsit5", "ude_inst_lv_s1_g5_w1_vern7", "ude_inst_lv_tanh32_g8_w1_tsit5", "ude_inst_lv_hudson_g8_w1_tsit5",
             "ude_inst_seir_ude_g64_w1_vern7", "ude_inst_seir_node_g64_w1_vern7", "ude_inst_kpp_ude_32_g32_w1_tsit5",
             "ude_inst_generic_7_g64_w1_tsit5", "ude_inst_generic_2_g64_w1_tsit5"]


def compile_plain(job):
    """`hipcc -c` and nothing else (see OBJ_NRW): no assembly text, no tool of tools/ between the compiler and the object"""
    src, obj, defs, log = job
    cmd = [HIPCC] + FLAGS + defs + ["-Rpass-analysis=kernel-resource-usage", "-MD", "-MF", obj + ".d", "-c", src, "-o", obj]
    stamp = obj + ".cmd"
    sig = hashlib.sha256(" ".join(cmd).encode()).hexdigest()
    if (os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_dep(obj + ".d"))
            and os.path.exists(stamp) and open(stamp).read() == sig):
        return obj, 0, "up to date"
    t_start = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    open(log, "w").write(r.stderr)
    if r.returncode == 0:
        os.utime(obj, (t_start, t_start))   # (see compile_one)
        open(stamp, "w").write(sig)
    return obj, r.returncode, r.stderr if r.returncode else ""


def kernel_work(objdir, extra):
    """the translation units of the shipping library as compile jobs into `objdir`, `extra` flags appended to each"""
    inst_src = os.path.join(CSRC, "ude_inst.hip")
    work = []
    for i in instances():
        defs = ["-DINST_NAME=" + i["name"], "-DINST_MODEL=" + i["model"], "-DINST_TAB=" + i["tab"], "-DINST_G=%d" % i["g"],
                "-DINST_VAR=%d" % i["w"], "-DINST_BLOCK=%d" % i["block"]] + i["extra"]
        work.append((inst_src, os.path.join(objdir, i["name"] + ".o"), defs + extra, os.path.join(objdir, i["name"] + ".log")))
    work.append((os.path.join(CSRC, "udecore.hip"), os.path.join(objdir, "udecore.o"), [] + extra, os.path.join(objdir, "udecore.log")))
    # the stochastic (deep-BSDE / LambaEM) path, SURVEY.md 8(f) N1 (UDE_HJB_DEFS: timing experiments)
    work.append((os.path.join(CSRC, "ude_hjb.hip"), os.path.join(objdir, "ude_hjb.o"), os.environ.get("UDE_HJB_DEFS", "").split() + extra, os.path.join(objdir, "ude_hjb.log")))
    # the lock-step matrix-core adjoint of the SEIR exposure UDE / the neural ODE
    work.append((os.path.join(CSRC, "ude_seir_ls.hip"), os.path.join(objdir, "ude_seir_ls.o"), [] + extra, os.path.join(objdir, "ude_seir_ls.log")))
    work.append((os.path.join(CSRC, "ude_node_ls.hip"), os.path.join(objdir, "ude_node_ls.o"), [] + extra, os.path.join(objdir, "ude_node_ls.log")))
    # the `fast` mode of the lock-step kernels: parameter cotangent as a block-level matrix-core accumulation
    work.append((os.path.join(CSRC, "ude_seir_lsf.hip"), os.path.join(objdir, "ude_seir_lsf.o"), [] + extra, os.path.join(objdir, "ude_seir_lsf.log")))
    work.append((os.path.join(CSRC, "ude_node_lsf.hip"), os.path.join(objdir, "ude_node_lsf.o"), [] + extra, os.path.join(objdir, "ude_node_lsf.log")))
    # the multi-GPU exchange step (RCCL bound with dlopen, one-shot P2P reducer), SURVEY.md 8(e)
    work.append((os.path.join(CSRC, "ude_comm.hip"), os.path.join(objdir, "ude_comm.o"), [] + extra, os.path.join(objdir, "ude_comm.log")))
    return work


def run_jobs(work, jobs, verbose):
    """compile `work` (slow units first) in `jobs` threads; the objects in the order of `work`"""
    # the translation units that take minutes (the 1024-point Fisher-KPP instances) start first
    slow = ("kpp_ude_1024_g64", "kpp_ude_1024", "kpp_ude_32", "kpp_s3", "ude_hjb", "ude_node_ls", "ude_seir_ls", "seir", "generic")
    order = sorted(range(len(work)), key=lambda i: next((k for k, pat in enumerate(slow) if pat in os.path.basename(work[i][1])), len(slow)))
    results = [None] * len(work)
    objs = []
    with ThreadPoolExecutor(jobs) as ex:
        for i, res in zip(order, ex.map(compile_one, [work[i] for i in order])):
            results[i] = res
        for obj, rc, msg in results:
            if verbose:
                print("  [%s] %s" % ("ok" if rc == 0 else "FAIL", os.path.relpath(obj, OBJ)), msg if rc else "")
            if rc:
                raise RuntimeError("hipcc failed for %s:\n%s" % (obj, msg))
            objs.append(obj)
    return objs


def build(verbose=True, jobs=None, ra2=False):
    os.makedirs(OBJ, exist_ok=True)
    write_instances_header()
    work = kernel_work(OBJ, [])
    # debug variants of the two host-side units (see LIB_DBG)
    write_poison_header()
    dbg = ["-DUDE_DEBUG_HOOKS=1", "-I" + OBJ]
    n_ship = len(work)
    work.append((os.path.join(CSRC, "udecore.hip"), os.path.join(OBJ, "udecore_dbg.o"), dbg, os.path.join(OBJ, "udecore_dbg.log")))
    work.append((os.path.join(CSRC, "ude_hjb.hip"), os.path.join(OBJ, "ude_hjb_dbg.o"), dbg, os.path.join(OBJ, "ude_hjb_dbg.log")))
    jobs = jobs or max(1, (os.cpu_count() or 2))
    objs = run_jobs(work, jobs, verbose)
    ship, dbg_objs = objs[:n_ship], objs[n_ship:]
    swap = {os.path.join(OBJ, "udecore.o"): dbg_objs[0], os.path.join(OBJ, "ude_hjb.o"): dbg_objs[1]}
    # the dynamic symbol table is include/udecore.h and nothing else (the instance getters, the lock-step getters and every C++
    # symbol stay local): a linker version script written from the header (tests/test_cabi_cpu.py compares `nm -D` with it)
    maps = {LIB: write_version_script(os.path.join(OBJ, "udecore.map"), []),
            LIB_DBG: write_version_script(os.path.join(OBJ, "udecore_dbg.map"), DBG_EXPORTS)}
    for lib, members in ((LIB, ship), (LIB_DBG, [swap.get(o, o) for o in ship])):
        if (not os.path.exists(lib)) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in members + [maps[lib]]):
            subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + maps[lib], "-o", lib] + members + ["-ldl"])
    # timing experiments (developer use): UDE_EXP_VARIANTS="name:file.hip:-DFOO=1 ..." builds <file>_<name>.o with the extra
    # define and links libudecore_<name>.so with that one object swapped (load it with UDE_LIB_VARIANT=<name>)
    for spec in os.environ.get("UDE_EXP_VARIANTS", "").split():
        name, fname, define = spec.split(":")
        base = fname[:-4]
        obj = os.path.join(OBJ, "%s_%s.o" % (base, name))
        o, rc, msg = compile_one((os.path.join(CSRC, fname), obj, define.split("#"), obj[:-2] + ".log"))
        if rc:
            raise RuntimeError("hipcc failed for %s:\n%s" % (obj, msg))
        members = [obj if os.path.basename(x) == base + ".o" else x for x in ship]
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + maps[LIB], "-o", os.path.join(HERE, "libudecore_%s.so" % name)] + members + ["-ldl"])
        if verbose:
            print("built experiment variant", name)
    # the second register allocation (see OBJ_RA2 above) -- every translation unit once more, minutes of compile time: built when asked
    # for (ra2 = True: __graft_entry__.build(), the driver's "does it build" step, so that tests/test_gpu_ra2.py finds it on the GPU box;
    # UDE_BUILD_RA2=1 for a developer), not by a plain `python -m universal_differential_equations_amd.build` (advisor, round 5)
    if (ra2 or os.environ.get("UDE_BUILD_RA2")) and not os.environ.get("UDE_SKIP_RA2"):
        os.makedirs(OBJ_RA2, exist_ok=True)
        ra2_objs = run_jobs(kernel_work(OBJ_RA2, RA2_FLAGS), jobs, verbose)
        if (not os.path.exists(LIB_RA2)) or any(os.path.getmtime(o) > os.path.getmtime(LIB_RA2) for o in ra2_objs + [maps[LIB]]):
            subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + maps[LIB], "-o", LIB_RA2] + ra2_objs + ["-ldl"])
    # the no-rewriter variant (see OBJ_NRW above): built with the second allocation, or UDE_BUILD_NRW=1
    if (ra2 or os.environ.get("UDE_BUILD_NRW")) and not os.environ.get("UDE_SKIP_NRW"):
        os.makedirs(OBJ_NRW, exist_ok=True)
        by_name = {os.path.basename(w[1])[:-2]: w for w in kernel_work(OBJ_NRW, NRW_FLAGS)}
        nrw = {}
        with ThreadPoolExecutor(jobs) as ex:
            for obj, rc, msg in ex.map(compile_plain, [by_name[n] for n in sorted(NRW_UNITS, key=lambda n: "kpp" not in n)]):
                if verbose:
                    print("  [%s] nrw/%s" % ("ok" if rc == 0 else "FAIL", os.path.basename(obj)), msg if rc else "")
                if rc:
                    raise RuntimeError("hipcc failed for %s:\n%s" % (obj, msg))
                nrw[os.path.basename(obj)] = obj
        members = [nrw.get(os.path.basename(o), o) for o in ship]
        assert sum(os.path.basename(o) in nrw for o in ship) == len(NRW_UNITS), "NRW_UNITS names a unit the shipping library does not have"
        if (not os.path.exists(LIB_NRW)) or any(os.path.getmtime(o) > os.path.getmtime(LIB_NRW) for o in members + [maps[LIB]]):
            subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + maps[LIB], "-o", LIB_NRW] + members + ["-ldl"])
    if verbose:
        print("built", LIB, "and", LIB_DBG)
    return LIB


if __name__ == "__main__":
    build(jobs=int(sys.argv[1]) if len(sys.argv) > 1 else None)

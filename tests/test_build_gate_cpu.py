"""The build-time gate against the code-generation defect of DESIGN.md 2a (register copies in front of the EXEC restore of a
join block: lanes that skipped the region keep a stale value).  tools/isa_endcf_fix.py repairs the device assembly of every
translation unit and refuses what it cannot prove safe; here: its behaviour on the patterns it must move, must leave alone and
must refuse, and -- when this container has built the library -- that every repaired assembly file under build/ is clean."""
import glob
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tool():
    spec = importlib.util.spec_from_file_location("isa_endcf_fix", os.path.join(ROOT, "tools", "isa_endcf_fix.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


KERNEL = "_Z6kernelv:\n"

# the site found in adj_kernel<LvUde<NetTanh32, 8>, Tsit5, fast> (round 4): three live-range copies in front of the join of a
# guarded loop; the lanes whose guard was false arrive with EXEC == 0 and would skip them
JOIN_WITH_COPIES = KERNEL + """\ts_and_saveexec_b64 s[12:13], s[4:5]
\ts_cbranch_execz .LBB0_3
; %bb.1:
\tv_add_f64 v[0:1], v[2:3], v[4:5]
.LBB0_3:
\tv_mov_b64_e32 v[142:143], v[18:19]
\tv_mov_b32_e32 v133, v12
\tv_mov_b64_e32 v[138:139], v[42:43]
\ts_or_b64 exec, exec, s[12:13]
\tv_add_f64 v[0:1], v[72:73], -v[70:71]
\ts_endpgm
"""


def test_copies_in_front_of_the_exec_restore_are_moved_behind_it():
    T = tool()
    text, rep = T.process(JOIN_WITH_COPIES)
    assert len(rep["fixed"]) == 1 and not rep["unhandled"]
    body = [l.strip() for l in text.split("\n") if l.strip()]
    i = body.index(".LBB0_3:")
    assert body[i + 1] == "s_or_b64 exec, exec, s[12:13]"
    assert body[i + 2:i + 5] == ["v_mov_b64_e32 v[142:143], v[18:19]", "v_mov_b32_e32 v133, v12", "v_mov_b64_e32 v[138:139], v[42:43]"]
    assert body[i + 5] == "s_nop 4"
    _, again = T.process(text, repair=False)
    assert not again["fixed"] and not again["unhandled"]       # idempotent: the repaired text is clean


def test_loop_exit_by_fall_through_is_an_entry_with_exec_zero():
    T = tool()
    src = KERNEL + """.LBB0_1:
\tv_add_u32_e32 v1, 1, v1
\ts_andn2_b64 exec, exec, s[2:3]
\ts_cbranch_execnz .LBB0_1
; %bb.2:
\tv_accvgpr_read_b32 v5, a7
\ts_or_b64 exec, exec, s[2:3]
\ts_endpgm
"""
    text, rep = T.process(src)
    assert len(rep["fixed"]) == 1
    body = [l.strip() for l in text.split("\n") if l.strip()]
    assert body.index("s_or_b64 exec, exec, s[2:3]") < body.index("v_accvgpr_read_b32 v5, a7")


def test_scalar_reload_of_the_mask_stays_in_front():
    """`v_readlane sN` (EXEC-independent) reloading the spilled mask must stay in front of the restore; a copy next to it moves"""
    T = tool()
    src = KERNEL + """\ts_cbranch_execz .LBB0_5
.LBB0_5:
\tv_readlane_b32 s2, v255, 47
\tv_mov_b32_e32 v9, v8
\tv_readlane_b32 s3, v255, 48
\ts_or_b64 exec, exec, s[2:3]
\ts_endpgm
"""
    text, rep = T.process(src)
    assert len(rep["fixed"]) == 1 and not rep["unhandled"]
    body = [l.strip() for l in text.split("\n") if l.strip()]
    i = body.index(".LBB0_5:")
    assert body[i + 1:i + 5] == ["v_readlane_b32 s2, v255, 47", "v_readlane_b32 s3, v255, 48", "s_or_b64 exec, exec, s[2:3]", "v_mov_b32_e32 v9, v8"]


def test_region_entry_after_a_wave_level_skip_is_left_alone():
    """EXEC == 0 on entry and the next EXEC write narrows further (a region ENTRY, plain or open-coded): the lanes stay off until an
    outer join; nothing in between can be owed to them"""
    T = tool()
    for entry in ("\ts_and_saveexec_b64 s[4:5], s[0:1]\n",
                  "\ts_mov_b64 s[8:9], exec\n\ts_and_b64 s[2:3], s[8:9], s[2:3]\n\ts_mov_b64 exec, s[2:3]\n"):
        src = KERNEL + "\ts_cbranch_execz .LBB0_58\n.LBB0_58:\n\tv_mov_b32_e32 v230, 0\n" + entry + "\ts_cbranch_execz .LBB0_60\n.LBB0_60:\n\ts_endpgm\n"
        text, rep = T.process(src)
        assert not rep["fixed"] and not rep["unhandled"] and text == src


def test_unprovable_moves_are_refused():
    T = tool()
    # the vector instruction writes the mask register the restore reads
    src = KERNEL + "\ts_cbranch_execz .LBB0_1\n.LBB0_1:\n\tv_cmp_lt_f64_e64 s[2:3], v[0:1], v[2:3]\n\ts_or_b64 exec, exec, s[2:3]\n\ts_endpgm\n"
    _, rep = T.process(src)
    assert rep["unhandled"] and not rep["fixed"]
    # a scalar instruction in the prefix depends on a vector result
    src = KERNEL + "\ts_cbranch_execz .LBB0_1\n.LBB0_1:\n\tv_cmp_lt_f64_e64 s[6:7], v[0:1], v[2:3]\n\ts_and_b64 s[8:9], s[6:7], s[10:11]\n\ts_or_b64 exec, exec, s[2:3]\n\ts_endpgm\n"
    _, rep = T.process(src)
    assert rep["unhandled"] and not rep["fixed"]
    # a memory instruction in front of a wait
    src = KERNEL + "\ts_cbranch_execz .LBB0_1\n.LBB0_1:\n\tds_read_b64 v[0:1], v2\n\ts_waitcnt lgkmcnt(0)\n\ts_or_b64 exec, exec, s[2:3]\n\ts_endpgm\n"
    _, rep = T.process(src)
    assert rep["unhandled"] and not rep["fixed"]


def test_only_pure_register_copies_may_move():
    """round 5 (advisor): arithmetic, v_cmp (SGPR mask result), MFMA, DPP and memory instructions in front of the restore are
    UNHANDLED -- the build fails -- even where the old register-disjointness test would have let them move; v_mov / v_accvgpr copies,
    plain encodings, inline constants included, still move"""
    T = tool()
    head = KERNEL + "\ts_cbranch_execz .LBB0_1\n.LBB0_1:\n"
    tail = "\ts_or_b64 exec, exec, s[2:3]\n\ts_endpgm\n"
    for bad in ("v_add_f64 v[0:1], v[2:3], v[4:5]", "v_cmp_lt_f64_e64 s[6:7], v[0:1], v[2:3]",
                "v_mfma_f64_16x16x4_f64 a[0:7], v[0:1], v[2:3], a[0:7]", "v_mov_b32_dpp v1, v2 row_shr:1 row_mask:0xf bank_mask:0xf",
                "v_mov_b32_sdwa v1, v2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD",
                "ds_write_b64 v8, v[66:67]", "global_load_dwordx2 v[0:1], v[4:5], off", "scratch_store_dword off, v3, off offset:16",
                "v_cndmask_b32_e32 v1, v2, v3, vcc", "v_mov_b32_e32 v1, s5"):
        _, rep = T.process(head + "\t" + bad + "\n" + tail)
        assert rep["unhandled"] and not rep["fixed"], bad
    for good in ("v_mov_b32_e32 v1, v2", "v_mov_b64_e32 v[0:1], v[2:3]", "v_accvgpr_write_b32 a3, v7", "v_accvgpr_read_b32 v7, a3",
                 "v_accvgpr_mov_b32 a1, a2", "v_mov_b32_e32 v1, 0", "v_mov_b32_e32 v1, 0x3ff00000"):
        _, rep = T.process(head + "\t" + good + "\n" + tail)
        assert rep["fixed"] and not rep["unhandled"], good


def test_vector_code_that_falls_through_a_label_into_a_restore_is_refused():
    """copies in front of ANOTHER block's label whose first instruction is the restore stand in front of a join all the same, but
    behind that label other predecessors arrive: not repairable, the build fails"""
    T = tool()
    src = KERNEL + "\ts_cbranch_execz .LBB0_1\n.LBB0_1:\n\tv_mov_b32_e32 v1, v2\n.LBB0_2:\n\ts_or_b64 exec, exec, s[2:3]\n\ts_endpgm\n"
    _, rep = T.process(src)
    assert rep["unhandled"] and not rep["fixed"]


def test_region_tail_that_falls_into_its_join_is_not_a_site():
    """an else-body that falls through into its own `s_or_b64 exec` (no label in between, entered with its lanes ON) is ordinary code"""
    T = tool()
    src = KERNEL + """\ts_or_saveexec_b64 s[2:3], s[36:37]
\ts_xor_b64 exec, exec, s[2:3]
\ts_cbranch_execz .LBB0_48
.LBB0_50:
\tv_add_u32_e32 v8, v254, v241
\tds_write2st64_b64 v8, v[66:67], v[68:69] offset1:1
\ts_or_b64 exec, exec, s[2:3]
.LBB0_48:
\ts_endpgm
"""
    text, rep = T.process(src)
    assert not rep["fixed"] and not rep["unhandled"] and text == src


def test_every_built_translation_unit_is_clean():
    """build.py keeps the repaired device assembly of every translation unit (build/*.fixed.s, this container only): none of
    them may contain a vector instruction between an EXEC == 0 entry and the EXEC restore"""
    files = sorted(glob.glob(os.path.join(ROOT, "universal_differential_equations_amd", "build", "*.fixed.s")) +
                   glob.glob(os.path.join(ROOT, "universal_differential_equations_amd", "build", "ra2", "*.fixed.s")))   # (+ the second register allocation)
    if not files:
        pytest.skip("no build/ directory here (the GPU box receives the built library only)")
    T = tool()
    for f in files:
        _, rep = T.process(open(f).read(), repair=False)
        assert not rep["fixed"] and not rep["unhandled"], "%s: %s" % (os.path.basename(f), (rep["fixed"] + rep["unhandled"])[:2])


# ---- the second pass of the pipeline: wait states in front of the hand-written DPP instructions (tools/isa_dpp_hazard.py) ----
def dpp_tool():
    spec = importlib.util.spec_from_file_location("isa_dpp_hazard", os.path.join(ROOT, "tools", "isa_dpp_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


DPP = "\t;;#ASMSTART\n\tv_fmac_f64_dpp v[34:35], v[104:105], v[100:101] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t;;#ASMEND\n"


def test_a_reload_of_the_dpp_operand_in_front_of_the_asm_statement_gets_its_wait_states():
    """the site of the first `-amdgpu-mfma-vgpr-form` build of the Fisher-KPP vector kernel: a weight register brought back from an
    AGPR directly in front of the v_fmac_f64_dpp that broadcasts out of it (VALU write -> DPP read: 2 wait states; the back end's
    hazard recogniser does not look inside an asm statement)"""
    T = dpp_tool()
    src = KERNEL + "\tv_accvgpr_read_b32 v104, a20\n\tv_accvgpr_read_b32 v105, a21\n" + DPP + "\ts_endpgm\n"
    text, rep = T.process(src)
    assert rep["dpp"] == 1 and len(rep["inserted"]) == 1 and rep["inserted"][0][1] == 2 and not rep["errors"]
    body = [l.strip() for l in text.split("\n") if l.strip()]
    assert body[body.index(";;#ASMSTART") - 1] == "s_nop 1"      # in front of the asm statement, not inside it
    assert not T.process(text, repair=False)[1]["inserted"]      # idempotent
    # one instruction in between: one more wait state is missing
    src = KERNEL + "\tv_accvgpr_read_b32 v105, a21\n\tv_add_u32_e32 v1, 1, v1\n" + DPP + "\ts_endpgm\n"
    text, rep = T.process(src)
    assert rep["inserted"][0][1] == 1 and "s_nop 0" in text
    # two instructions (or an s_nop 1) in between: nothing to do; a write of ANOTHER register: nothing to do
    for mid in ("\tv_add_u32_e32 v1, 1, v1\n\tv_add_u32_e32 v2, 1, v2\n", "\ts_nop 1\n"):
        text, rep = T.process(KERNEL + "\tv_accvgpr_read_b32 v105, a21\n" + mid + DPP + "\ts_endpgm\n")
        assert not rep["inserted"]
    text, rep = T.process(KERNEL + "\tv_accvgpr_read_b32 v100, a21\n\tv_mov_b64_e32 v[34:35], 0\n" + DPP + "\ts_endpgm\n")
    assert not rep["inserted"]                                   # (destination and second source are ordinary operands)


def test_a_block_boundary_or_an_exec_write_in_the_window_is_padded_and_other_dpp_controls_are_refused():
    T = dpp_tool()
    text, rep = T.process(KERNEL + "\tv_add_u32_e32 v1, 1, v1\n.LBB0_7:\n" + DPP + "\ts_endpgm\n")
    assert rep["inserted"] and rep["inserted"][0][1] == 2         # another predecessor could end in anything
    text, rep = T.process(KERNEL + "\tv_cmpx_gt_f64_e32 v[0:1], v[2:3]\n\tv_add_u32_e32 v1, 1, v1\n" + DPP + "\ts_endpgm\n")
    assert rep["inserted"][0][1] == 4                             # VALU write of EXEC -> DPP: 5 wait states, one instruction is in between
    bad = KERNEL + "\t;;#ASMSTART\n\tv_fmac_f64_dpp v[0:1], v[2:3], v[4:5] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t;;#ASMEND\n"
    assert T.process(bad)[1]["errors"]
    # DPP instructions the compiler selected itself (outside an asm statement) are its own business
    own = KERNEL + "\tv_mov_b32_e32 v3, v9\n\tv_mov_b32_dpp v2, v3 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_endpgm\n"
    assert T.process(own)[1]["dpp"] == 0


def test_every_built_translation_unit_has_its_dpp_wait_states():
    files = sorted(glob.glob(os.path.join(ROOT, "universal_differential_equations_amd", "build", "*.fixed.s")) +
                   glob.glob(os.path.join(ROOT, "universal_differential_equations_amd", "build", "ra2", "*.fixed.s")))
    if not files:
        pytest.skip("no build/ directory here (the GPU box receives the built library only)")
    T = dpp_tool()
    seen = 0
    for f in files:
        _, rep = T.process(open(f).read(), repair=False)
        assert not rep["inserted"] and not rep["errors"], "%s: %s" % (os.path.basename(f), (rep["inserted"] + rep["errors"])[:2])
        seen += rep["dpp"]
    assert seen > 0   # (the Fisher-KPP vector kernel is made of them)


def test_the_no_rewriter_objects_never_saw_the_rewriter():
    """libudecore_nrw.so's swapped-in objects (build.py: OBJ_NRW; tests/test_gpu_ra2.py runs the oracle-comparing files against them) are
    what a plain `hipcc -c -O0` wrote: no assembly text in their directory, no line of either assembly tool in their logs, and the
    structural work-around is visible -- without live-range splitting the long-lived values of these kernels live in scratch"""
    objdir = os.path.join(ROOT, "universal_differential_equations_amd", "build", "nrw")
    logs = glob.glob(os.path.join(objdir, "*.log"))
    if not logs:
        pytest.skip("the no-rewriter variant has not been built (UDE_BUILD_NRW=1 / __graft_entry__.build())")
    import importlib.util
    spec = importlib.util.spec_from_file_location("ude_build", os.path.join(ROOT, "universal_differential_equations_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.NRW_FLAGS == ["-O0"]
    assert sorted(os.path.basename(x)[:-4] for x in logs) == sorted(b.NRW_UNITS)
    assert not glob.glob(os.path.join(objdir, "*.s")), "no assembly text may exist next to objects that come out of `hipcc -c`"
    for lg in logs:
        txt = open(lg).read()
        assert "endcf-fix" not in txt and "dpp-hazard" not in txt, lg
    scratch = [int(x) for lg in logs for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", open(lg).read())]
    assert scratch and max(scratch) > 1000

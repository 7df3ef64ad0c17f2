"""DESIGN.md's register / scratch / occupancy numbers are the build's: profiles/r06_kernel_resources.md is generated from the
`-Rpass-analysis=kernel-resource-usage` remarks of the compiles that produced the linked objects (tools/kernel_resources.py), and
where this container holds a build/ directory the committed table must equal a fresh one (the GPU box receives the library only)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tool():
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_resource_table_is_the_one_the_build_logs_give():
    T = tool()
    if not os.path.isdir(T.BUILD) or not os.path.exists(os.path.join(T.BUILD, "ude_seir_ls.log")):
        pytest.skip("no build/ directory here")
    assert os.path.exists(T.TABLE), "profiles/r06_kernel_resources.md is missing: tools/kernel_resources.py --table --write"
    fresh = T.table()
    assert "(not built)" not in fresh and "(no kernel matching" not in fresh, fresh
    assert open(T.TABLE).read() == fresh, "the committed table is stale: tools/kernel_resources.py --table --write"


def test_design_md_cites_the_generated_table_and_types_no_scratch_numbers():
    txt = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert "r06_kernel_resources.md" in txt

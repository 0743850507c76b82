"""The generated headers that are committed (csrc/umpc_gen.h, umpc_admm_asm.h, umpc_admm_asm64.h, umpc_step_asm.h) must be
what the generators in the tree produce: build() rewrites them, so a stale committed copy would only show up as a diff."""
import os

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robobee3d_amd", "csrc")


@pytest.mark.parametrize("module, fname", [("codegen", "umpc_gen.h"), ("asmgen", "umpc_admm_asm.h"),
                                           ("asmgen64", "umpc_admm_asm64.h"), ("asmstep", "umpc_step_asm.h"),
                                           ("asmstep", "umpc_step_asm_quad.h"), ("codegen_n3", "umpc_n3_general.h"),
                                           ("asmgen64", "umpc_admm_asm64_quad.h")])
def test_committed_header_is_current(tmp_path, module, fname):
    import importlib
    from robobee3d_amd import asmgen
    assert not asmgen.generator_switches(), "generator switches must be off for this comparison"
    mod = importlib.import_module("robobee3d_amd." + module)
    out = str(tmp_path / fname)
    if fname == "umpc_admm_asm64_quad.h":
        mod.write_quad(out)
        assert open(str(tmp_path / "umpc_quad64_tab.h")).read() == open(os.path.join(CSRC, "umpc_quad64_tab.h")).read()
    elif fname.endswith("_quad.h"):
        mod.write(out, quad=True)
    else:
        mod.write(out)
    assert open(out).read() == open(os.path.join(CSRC, fname)).read(), "%s is stale: run __graft_entry__.build()" % fname


def test_committed_headers_say_which_switches_made_them():
    """every committed instruction stream carries the switch banner of a default-environment generation"""
    for fname in ("umpc_admm_asm.h", "umpc_admm_asm64.h", "umpc_step_asm.h", "umpc_step_asm_quad.h"):
        head = open(os.path.join(CSRC, fname)).read(400)
        assert "// generator switches: none (defaults = the shipped kernels)" in head, fname


def test_build_refuses_generator_switches(monkeypatch):
    """_lib.build() rewrites tracked headers and the shipped library, so it must refuse to run under a stray
    A/B switch (variants go through tools/build_variant.py into robobee3d_amd/variants/)."""
    from robobee3d_amd import _lib, asmgen
    monkeypatch.setenv("UMPC_ASM_LIMIT_FAST", "1")
    assert asmgen.generator_switches() == {"UMPC_ASM_LIMIT_FAST": "1"}
    assert "UMPC_ASM_LIMIT_FAST=1" in asmgen.switch_banner()
    before = {f: os.path.getmtime(os.path.join(CSRC, f)) for f in ("umpc_step_asm.h", "umpc_admm_asm.h")}
    with pytest.raises(RuntimeError, match="generator switches"):
        _lib.build()
    assert before == {f: os.path.getmtime(os.path.join(CSRC, f)) for f in before}
    monkeypatch.setenv("UMPC_QP_KERNEL", "tables")       # a run-time diagnostic, not a generator switch
    monkeypatch.delenv("UMPC_ASM_LIMIT_FAST")
    assert asmgen.generator_switches() == {}


def test_run_time_variables_of_the_host_code_are_not_generator_switches(monkeypatch):
    """generator_switches() matches by prefix, so a RUN-time variable of the host code whose name starts like a
    generator switch (UMPC_ASM_SKEW_US, UMPC_QP_NO_ASM ...) must be listed in NOT_SWITCHES, or setting it makes build()
    refuse and the header test fail. Every getenv("UMPC_...") of csrc/ is checked against the list."""
    import glob
    import re
    from robobee3d_amd import asmgen
    names = set()
    for path in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")):
        names |= set(re.findall(r'getenv\("(UMPC_[A-Z0-9_]+)"\)', open(path).read()))
    assert {"UMPC_ASM_SKEW_US", "UMPC_QP_NO_ASM", "UMPC_QUAD"} <= names
    for n in sorted(names):
        if n.startswith(asmgen.SWITCH_PREFIXES):
            assert n in asmgen.NOT_SWITCHES, "%s is read at run time by csrc/ but counts as a generator switch" % n
        monkeypatch.setenv(n, "1")
    assert asmgen.generator_switches() == {}

"""The generated headers that are committed (csrc/umpc_gen.h, umpc_admm_asm.h, umpc_admm_asm64.h, umpc_step_asm.h) must be
what the generators in the tree produce: build() rewrites them, so a stale committed copy would only show up as a diff."""
import os

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robobee3d_amd", "csrc")


@pytest.mark.parametrize("module, fname", [("codegen", "umpc_gen.h"), ("asmgen", "umpc_admm_asm.h"),
                                           ("asmgen64", "umpc_admm_asm64.h"), ("asmstep", "umpc_step_asm.h")])
def test_committed_header_is_current(tmp_path, module, fname):
    import importlib
    for k in ("UMPC_ASM64_TIMING", "UMPC_ASM64_AHEAD", "UMPC_ASM64_MERGE", "UMPC_ASM_XV", "UMPC_ASM_LIMIT_FAST"):
        assert k not in os.environ, "generator switches must be off for this comparison"
    mod = importlib.import_module("robobee3d_amd." + module)
    out = str(tmp_path / fname)
    mod.write(out)
    assert open(out).read() == open(os.path.join(CSRC, fname)).read(), "%s is stale: run __graft_entry__.build()" % fname

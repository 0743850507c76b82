#!/usr/bin/env python3
"""Builds robobee3d_amd/variants/libumpc_<name>.so from the current sources with generator switches taken from the
environment (UMPC_ASM_*), for A/B timing of step-kernel variants inside ONE gpurun call (select with UMPC_LIB).
usage: UMPC_ASM_XV=0 tools/build_variant.py noxv
       UMPC_VARIANT_DEFS="-DUMPC_SCALING_ITERS=1" tools/build_variant.py ruiz1     (extra compiler flags, timing diagnostics)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robobee3d_amd import _lib, asmgen, asmstep  # noqa: E402

name = sys.argv[1]
vdir = os.path.join(ROOT, "robobee3d_amd", "variants")
os.makedirs(vdir, exist_ok=True)
# everything else up to date (objects of the other units are reused): the default build, in a child process WITHOUT the
# generator switches -- _lib.build() refuses to regenerate the shipped kernels under them
clean = {k: v for k, v in os.environ.items() if k not in asmgen.generator_switches()}
subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from robobee3d_amd import _lib; _lib.build()" % ROOT],
               check=True, env=clean)
hdr = os.path.join(vdir, "umpc_step_asm.h")
asmstep.write(hdr)
obj = os.path.join(vdir, name + ".o")
csrc = os.path.join(ROOT, "robobee3d_amd", "csrc")
subprocess.run(["hipcc"] + _lib.HIPCC_FLAGS + os.environ.get("UMPC_VARIANT_DEFS", "").split() + ["-DUMPC_STEP_ASM_HEADER=\"%s\"" % hdr, "-c", "-o", obj, _lib.SRC], check=True, cwd=csrc)
objs = [os.path.join(_lib.OBJ_DIR, f) for f in os.listdir(_lib.OBJ_DIR) if f.endswith(".o") and f != "umpc_mi355x.hip.o"]
out = os.path.join(vdir, "libumpc_%s.so" % name)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + objs, check=True)
os.remove(obj)
print(out)

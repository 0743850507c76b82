"""ctypes loader for libumpc_mi355x.so (the C ABI in include/umpc_mi355x.h).

There is NO fallback: if the shared library is missing this raises, and the
library itself refuses to create a controller when no HIP device is present.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO_PATH = os.path.join(HERE, "libumpc_mi355x.so")
SRC = os.path.join(HERE, "csrc", "umpc_mi355x.hip")
SRC_BQP = os.path.join(HERE, "csrc", "umpc_bqp.hip")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17"]

UMPC_F32, UMPC_F64 = 0, 1
STATE_ROWS, CTRL_ROWS, REF_ROWS, OUT_ROWS, STAT_ROWS = 18, 127, 9, 9, 2
NX, NC, NADATA = 45, 39, 48

# every symbol include/umpc_mi355x.h declares
EXPORTS = ["umpcInit", "umpcUpdate", "umpcS", "umpcLastStatus", "umpcRelease", "umpcLiveControllers", "umpcSetCompat",
           "umpcBatchDefaultParams", "umpcBatchCreate", "umpcBatchDestroy", "umpcBatchInitCtrl",
           "umpcBatchRollout", "umpcBatchUpdate", "umpcBatchPlant", "umpcBatchAssemble",
           "umpcBatchSize", "umpcBatchDtype", "umpcAxIdx", "umpcKKTPerm", "umpcNnzL",
           "umpcBatchSetTask", "umpcBatchTime", "umpcBatchSetWeights", "umpcBatchSetStepKernel", "umpcBatchSetGlobalBatch", "umpcBatchGlobalBatch", "umpcBatchReactive", "umpcBatchTaskReference",
           "umpcLastError", "umpcKernelName", "umpcBatchKernelName", "wlConInit", "wlConUpdate", "wlconS", "umpcBatchWLUpdate", "umpcBatchSetWL", "umpcBatchModel",
           "umpcQPDefaultSettings", "umpcQPCreate", "umpcQPDestroy", "umpcQPSetMaxIter", "umpcQPSetCheckTermination", "umpcQPSetAdaptiveRho", "umpcQPUseTables", "umpcQPSetKernel", "umpcQPKernelName", "umpcQPSolve", "umpcQPGather", "umpcQPGatherUpdate",
           "umpcP5fStep", "umpcP5fStepU", "umpcP5fLinearise", "umpcP5fTick", "umpcNAssemble", "umpcNExtract"]


class NParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("dt", "g", "TtoWmax", "ws", "wds", "wpr", "wpf", "wvr", "wvf", "wthrust",
                                          "wmom")] + [("Ib", C.c_double * 3)]


class QPSettings(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "eps_prim_inf",
                                          "eps_dual_inf")] + [("max_iter", C.c_int), ("scaling", C.c_int),
                                                              ("check_termination", C.c_int), ("adaptive_rho_interval", C.c_int)]


class FunApprox_t(C.Structure):
    # funapprox.h:18-23
    _fields_ = [("k", C.c_int), ("a0", C.c_float), ("a1", C.c_float * 4), ("A2", C.c_float * 16)]


class WLCon_t(C.Structure):
    # funapprox.h:37-41
    _fields_ = [("u0", C.c_float * 4), ("umin", C.c_float * 4), ("umax", C.c_float * 4), ("dumax", C.c_float * 4),
                ("Qw", C.c_float * 36), ("fa", FunApprox_t * 6)]


class BatchParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("dt", "g", "TtoWmax", "ws", "wds", "wpr", "wpf", "wvr",
                                          "wvf", "wthrust", "wmom")] + \
               [("Ib", C.c_double * 3), ("maxIter", C.c_int), ("dtsim", C.c_double),
                ("taulim", C.c_double), ("nsub", C.c_int), ("plant_mode", C.c_int)]


class UprightMPC_t(C.Structure):
    # include/umpc_mi355x.h == template/uprightmpc2/uprightmpc2.h:27-43
    _fields_ = [
        ("dt", C.c_float), ("g", C.c_float), ("Tmax", C.c_float),
        ("Qyr", C.c_float * 6), ("Qyf", C.c_float * 6), ("Qdyr", C.c_float * 6),
        ("Qdyf", C.c_float * 6), ("R", C.c_float * 3), ("smin", C.c_float * 3),
        ("smax", C.c_float * 3), ("e3h", C.c_float * 9), ("e3hIbi", C.c_float * 9),
        ("l", C.c_float * NC), ("u", C.c_float * NC), ("q", C.c_float * NX),
        ("Px_data", C.c_float * NX), ("Ax_data", C.c_float * NADATA),
        ("Ax_idx", C.c_int * NADATA), ("nAxT0dt", C.c_int), ("nAxdt", C.c_int),
        ("c0", C.c_float * 6), ("T0", C.c_float),
    ]


RESOURCE_LIMITS = os.path.join(HERE, "csrc", "resource_limits.json")


def _check_resources(remarks):
    """Parses hipcc's -Rpass-analysis=kernel-resource-usage remarks of the step-kernel translation unit and fails
    the build when a kernel's register / scratch / LDS outcome is no longer the measured one (csrc/resource_limits.json):
    the fp32 step kernel hands registers v0/v1/s[4:11] to a generated assembly block that clobbers v2-v245 and every
    AGPR, so a compiler change that grows the scratch frame or drops the 512-register allocation must be seen here,
    not as a silent slowdown on the GPU."""
    import json
    import re
    res, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    _validate_resources(res)
    # persisted next to the object: a later build that REUSES the object validates this record again
    with open(RESOURCES_JSON, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    return res


RESOURCES_JSON = os.path.join(OBJ_DIR, "resources.json")


def _validate_resources(res):
    import json
    if not os.path.exists(RESOURCE_LIMITS):
        return res
    limits = json.load(open(RESOURCE_LIMITS))
    for pat, lim in limits.items():
        hits = [k for k in res if pat in k]
        if not hits:
            raise RuntimeError("resource check: no kernel matches %r" % pat)
        for k in hits:
            r = res[k]
            for key, op in (("VGPRs", "=="), ("AGPRs", "=="), ("LDS", "=="), ("ScratchSize", "<=")):
                if key in lim and key not in r:     # a remark the compiler no longer prints is a failed check, not a KeyError
                    raise RuntimeError("resource check: %s missing for %s (hipcc remark format changed?)" % (key, k))
                if key in lim and not (r[key] == lim[key] if op == "==" else r[key] <= lim[key]):
                    raise RuntimeError("resource check failed for %s: %s = %d, recorded %s %d (csrc/resource_limits.json)"
                                       % (k, key, r[key], op, lim[key]))
    return res


def build(force=False, verbose=False):
    """Generate umpc_gen.h / umpc_admm_asm.h and compile the HIP library for gfx950 (works without a GPU).
    One object per translation unit (recompiled only when it or its headers changed), then one link."""
    from . import asmgen, asmgen64, asmstep, codegen, codegen_n3, codegen_qp
    sw = asmgen.generator_switches()
    # (a scratch COPY of the tree may be built under switches for A/B experiments: UMPC_VARIANT_ROOT must name exactly the
    # root being built, so the variable cannot unlock the tree it was not meant for; tools/build_qp_variant.sh)
    if sw and os.environ.get("UMPC_VARIANT_ROOT") != ROOT:
        # build() REWRITES the tracked generated headers and the shipped .so: a stray A/B switch in a test, bench or
        # profile shell must not change the kernels silently (one of them, UMPC_ASM_LIMIT_FAST, changes the numerics).
        # Variants are built by tools/build_variant.py into robobee3d_amd/variants/ and selected with UMPC_LIB.
        raise RuntimeError("generator switches are set in the environment (%s): refusing to regenerate the shipped kernels; "
                           "unset them, or build a variant with tools/build_variant.py" % " ".join("%s=%s" % kv for kv in sw.items()))
    gen, _ = codegen.write()
    gn3 = codegen_n3.write()
    gasm, _ = asmgen.write()
    gasm64 = asmgen64.write()[0]
    gasm64q = asmgen64.write_quad()[0]
    gstep, _ = asmstep.write()
    gquad, _ = asmstep.write(quad=True)
    greg, gqp_units = codegen_qp.write()
    hdr = os.path.join(ROOT, "include", "umpc_mi355x.h")
    csrc = os.path.join(HERE, "csrc")
    units = [(SRC, [gen, gasm, gasm64, gasm64q, gstep, gquad, gn3, hdr] + [os.path.join(csrc, f) for f in ("umpc_step.h", "umpc_models.h", "umpc_err.h")]),
             (SRC_BQP, [hdr, greg, os.path.join(csrc, "umpc_bqp_common.h"), os.path.join(csrc, "umpc_err.h")])]
    gen_hdrs = [os.path.join(csrc, "gen", f) for f in os.listdir(os.path.join(csrc, "gen")) if f.endswith(".h")]
    units += [(u, [os.path.join(csrc, "umpc_bqp_common.h")] + gen_hdrs) for u in gqp_units]
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs, relink = [], force or not os.path.exists(SO_PATH)
    todo = []
    for src, deps in units:
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if (force or not os.path.exists(obj)
                or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in [src] + deps)):
            extra = ["-Rpass-analysis=kernel-resource-usage"] if src == SRC else []
            todo.append(["hipcc"] + HIPCC_FLAGS + extra + ["-c", "-o", obj, src])
            relink = True
    step_obj = os.path.join(OBJ_DIR, os.path.basename(SRC) + ".o")
    if not any(c[-1] == SRC for c in todo):
        # the step-kernel object is reused: its recorded resource usage must still pass (and must exist)
        import json
        try:
            _validate_resources(json.load(open(RESOURCES_JSON)))
        except Exception:
            todo.insert(0, ["hipcc"] + HIPCC_FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", step_obj, SRC])
            relink = True
    for stale in set(os.listdir(OBJ_DIR)) - {os.path.basename(o) for o in objs} - {os.path.basename(RESOURCES_JSON)}:
        os.remove(os.path.join(OBJ_DIR, stale))
    # one hipcc per translation unit, at most one per host CPU at a time
    running, width = [], max(1, min(len(todo), os.cpu_count() or 1))
    while todo or running:
        while todo and len(running) < width:
            cmd = todo.pop(0)
            if verbose:
                print(" ".join(cmd))
            running.append((cmd, subprocess.Popen(cmd, cwd=csrc, stderr=subprocess.PIPE)))
        cmd, p = running.pop(0)
        _, err = p.communicate()
        if p.returncode != 0:
            for _, q in running:
                q.kill()
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), (err or b"").decode()[-4000:]))
        if cmd[-1] == SRC:
            try:
                _check_resources((err or b"").decode())
            except Exception:
                os.remove(cmd[-2])     # the object must not survive a failed check (it would be reused unchecked)
                raise
    if relink or any(os.path.getmtime(SO_PATH) < os.path.getmtime(o) for o in objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=csrc, stderr=None if verbose else subprocess.DEVNULL)
    build_ext(verbose=verbose)
    return SO_PATH


EXT_SRC = os.path.join(HERE, "csrc", "uprightmpc2py_ext.cpp")


def ext_path():
    import sysconfig
    return os.path.join(HERE, "_uprightmpc2py" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_ext(force=False, verbose=False):
    """The compiled Python module of the drop-in boundary (csrc/uprightmpc2py_ext.cpp: pybind11 over the C ABI, the
    counterpart of template/uprightmpc2/py/uprightmpc2py.cpp): host compiler only, linked against libumpc_mi355x.so next to
    it (rpath $ORIGIN). In-tree, like the library, so that it travels to the GPU box."""
    import sysconfig
    import pybind11
    out = ext_path()
    hdr = os.path.join(ROOT, "include", "umpc_mi355x.h")
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in (EXT_SRC, hdr, SO_PATH)):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-I", pybind11.get_include(),
           "-I", sysconfig.get_paths()["include"], "-I", os.path.join(ROOT, "include"), EXT_SRC, "-o", out,
           "-L", HERE, "-l:libumpc_mi355x.so", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError("libumpc_mi355x.so is not built (run `python -c 'import __graft_entry__ as g; "
                               "g.build()'`); robobee3d_amd has no CPU fallback")
        # UMPC_LIB: another build of the same library (A/B timing of kernel variants on ONE box; devices differ by
        # several per cent, so variants are only comparable inside one gpurun call)
        L = C.CDLL(os.environ.get("UMPC_LIB") or SO_PATH)
        L.umpcBatchCreate.restype = C.c_void_p
        L.umpcBatchCreate.argtypes = [C.POINTER(BatchParams), C.c_int, C.c_int]
        L.umpcBatchDestroy.argtypes = [C.c_void_p]
        L.umpcBatchInitCtrl.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.umpcBatchRollout.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 11
        L.umpcBatchUpdate.argtypes = [C.c_void_p] + [C.c_void_p] * 9
        L.umpcBatchReactive.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8
        L.umpcBatchTaskReference.argtypes = [C.c_void_p, C.c_double] + [C.c_void_p] * 3
        L.umpcBatchPlant.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.umpcBatchAssemble.argtypes = [C.c_void_p] + [C.c_void_p] * 10
        L.umpcLastError.restype = C.c_char_p
        L.umpcKernelName.restype = C.c_char_p
        L.umpcBatchKernelName.restype = C.c_char_p
        L.umpcBatchKernelName.argtypes = [C.c_void_p]
        L.umpcAxIdx.restype = C.POINTER(C.c_int * NADATA)
        L.umpcKKTPerm.restype = C.POINTER(C.c_int * (NX + NC))
        L.umpcBatchSetTask.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double]
        L.umpcBatchSetWeights.argtypes = [C.c_void_p, C.c_void_p]
        L.umpcBatchSetStepKernel.argtypes = [C.c_void_p, C.c_int]
        L.umpcBatchSetGlobalBatch.argtypes = [C.c_void_p, C.c_longlong]
        L.umpcBatchGlobalBatch.argtypes = [C.c_void_p]
        L.umpcBatchGlobalBatch.restype = C.c_longlong
        L.umpcBatchTime.argtypes = [C.c_void_p]
        L.umpcBatchTime.restype = C.c_double
        L.umpcBatchModel.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double] + [C.c_void_p] * 4
        L.umpcBatchWLUpdate.argtypes =[C.POINTER(WLCon_t), C.c_int, C.c_int] + [C.c_void_p] * 5
        L.umpcBatchSetWL.argtypes = [C.c_void_p, C.POINTER(WLCon_t), C.POINTER(C.c_double), C.c_void_p, C.c_void_p]
        L.umpcUpdate.restype = C.c_int
        L.umpcQPDefaultSettings.argtypes = [C.POINTER(QPSettings)]
        L.umpcQPCreate.restype = C.c_void_p
        L.umpcQPCreate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(QPSettings)]
        L.umpcQPDestroy.argtypes = [C.c_void_p]
        L.umpcQPSetMaxIter.argtypes = [C.c_void_p, C.c_int]
        L.umpcQPSetCheckTermination.argtypes = [C.c_void_p, C.c_int]
        L.umpcQPSetAdaptiveRho.argtypes = [C.c_void_p, C.c_int]
        L.umpcQPUseTables.argtypes = [C.c_void_p, C.c_int]
        L.umpcQPSetKernel.argtypes = [C.c_void_p, C.c_int]
        L.umpcQPKernelName.argtypes = [C.c_void_p]
        L.umpcQPKernelName.restype = C.c_char_p
        L.umpcQPSolve.argtypes = [C.c_void_p] * 15
        L.umpcQPGather.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.umpcQPGatherUpdate.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.umpcP5fStep.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double] + [C.c_void_p] * 4
        L.umpcP5fStepU.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double] + [C.c_void_p] * 3
        L.umpcP5fLinearise.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.umpcP5fTick.argtypes = [C.c_void_p] * 14 + [C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.umpcNAssemble.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(NParams)] + [C.c_void_p] * 10
        L.umpcNExtract.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double] + [C.c_void_p] * 5
        L.umpcLastStatus.restype = C.c_int
        L.umpcSetCompat.restype = C.c_int
        _lib = L
    return _lib


def default_params():
    p = BatchParams()
    lib().umpcBatchDefaultParams(C.byref(p))
    return p

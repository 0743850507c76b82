"""CPU: the C-ABI library loads and exports every symbol include/umpc_mi355x.h declares
(no compute calls without a GPU), refuses to create a controller without a device, and the
host-side constants agree with the header."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from robobee3d_amd import _lib
    _lib.build()
    return _lib


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "umpc_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(umpc[A-Z]\w*|wlCon[A-Z]\w*|wlconS)\s*\(", hdr))
    assert {"umpcInit", "umpcUpdate", "umpcS"} <= declared and len(declared) >= 20
    L = C.CDLL(lib.SO_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(lib.EXPORTS)


def test_reference_struct_layout_and_constants(lib):
    # template/uprightmpc2/uprightmpc2.h:27-43 is 1308 bytes; vectors()/matrices() read fields by offset
    assert C.sizeof(lib.UprightMPC_t) == 1308
    assert lib.UprightMPC_t.l.offset == 4 * (3 + 27 + 6 + 18)
    hdr = open(os.path.join(ROOT, "include", "umpc_mi355x.h")).read()
    for name, val in (("UMPC_STATE_ROWS", lib.STATE_ROWS), ("UMPC_CTRL_ROWS", lib.CTRL_ROWS),
                      ("UMPC_REF_ROWS", lib.REF_ROWS), ("UMPC_OUT_ROWS", lib.OUT_ROWS)):
        assert int(re.search(r"#define %s (\d+)" % name, hdr).group(1)) == val


def test_static_tables_without_gpu(lib, structure):
    L = lib.lib()
    assert np.array_equal(np.array(L.umpcAxIdx().contents), structure["Ax_idx"])  # uprightmpc2.c:65-113
    perm = np.array(L.umpcKKTPerm().contents)
    assert sorted(perm.tolist()) == list(range(84))
    assert L.umpcNnzL() == 213          # same fill as the reference's AMD ordering (workspace.c:1260)
    p = lib.default_params()
    assert (p.dt, p.maxIter, p.nsub, p.dtsim, p.taulim) == (5.0, 50, 25, 0.2, 100.0)


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = lib.lib()
    p = lib.default_params()
    h = L.umpcBatchCreate(C.byref(p), 64, 0)
    assert not h and b"no HIP device" in L.umpcLastError()
    from robobee3d_amd.batch import BatchUprightMPC
    with pytest.raises(RuntimeError):
        BatchUprightMPC(64)


def test_symbolic_matches_reference_tables(structure):
    from robobee3d_amd import symbolic
    s = symbolic.analyse(3, perm=structure["perm"])
    for k in "A_p A_i K_p K_i PtoKKT AtoKKT rhotoKKT etree Lnz L_p L_i Ax_idx".split():
        assert np.array_equal(np.array(getattr(s, k)), structure[k]), k
    own = symbolic.analyse(3)
    assert len(own.L_i) == 213 and sorted(own.perm) == list(range(84))


def test_codegen_is_deterministic():
    from robobee3d_amd import asmgen, codegen
    assert codegen.emit()[0] == codegen.emit()[0]
    a, _ = asmgen.program()
    b, _ = asmgen.program()
    assert a == b


_IMPORT_LINES = """
import sys
sys.path[:] = [p for p in sys.path if p not in ("", %(root)r)] + [%(root)r]
from uprightmpc2py import UprightMPC2C # C version               (template/template_controllers.py:5, verbatim)
from uprightmpc2py import UprightMPC2C, WLCon                    # (template/robobee_test_controllers.py:9, verbatim)
import inspect, uprightmpc2py, os
assert os.path.dirname(os.path.abspath(uprightmpc2py.__file__)) == %(root)r
import re
def params(f):
    # a compiled (pybind11) method carries its signature in the first line of its docstring, a Python one in inspect
    try:
        sig = inspect.signature(f)
        return list(sig.parameters)[1:], {k: v.default for k, v in sig.parameters.items() if v.default is not inspect._empty}
    except ValueError:
        head = f.__doc__.splitlines()[0]
        inner = head[head.index("(") + 1:head.rindex(") ->")]
        # split on the commas that precede "name:" (annotations contain commas of their own)
        parts = re.split(r", (?=[A-Za-z_0-9]+: )", inner)[1:]
        names = [p_.split(":")[0] for p_ in parts]
        dflt = {p_.split(":")[0]: float(p_.rsplit("=", 1)[1]) for p_ in parts if "=" in p_.rsplit("]", 1)[-1]}
        return names, dflt
# template/uprightmpc2.py:139 calls update with SIX arguments; template/uprightmpc2/py/uprightmpc2py.cpp:38 has seven
names, dflt = params(UprightMPC2C.update)
assert names == ["p0", "R0", "dq0", "pdes", "dpdes", "sdes", "actualT0"], names
assert dflt == {"actualT0": -1.0}, dflt
assert params(UprightMPC2C.__init__)[0] == ["dt", "g", "TtoWmax", "ws", "wds", "wpr", "wpf", "wvr", "wvf", "wthrust", "wmom",
                                            "Ib", "maxIter"]
assert params(WLCon.__init__)[0] == ["u0", "umin", "umax", "dumax", "Qw", "controlRate", "popts"]
print("ok")
"""


def test_reference_import_lines_run_unchanged(lib):
    """`from uprightmpc2py import UprightMPC2C, WLCon` -- the import the reference executes -- works
    with only the repository root added to sys.path (a top-level module, not a one-line edit of the reference)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", _IMPORT_LINES % dict(root=ROOT)], capture_output=True, text=True,
                       cwd="/", env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]


def test_compiled_python_module_is_built_and_mirrors_the_reference_classes(lib):
    """The counterpart of the reference's pybind11 extension (template/uprightmpc2/py/uprightmpc2py.cpp:70-80: classes
    UprightMPC2C(13 arguments).update / vectors / matrices and WLCon(7 arguments).update) exists as a COMPILED module
    (csrc/uprightmpc2py_ext.cpp, built in-tree by _lib.build()), links the library next to it, and is what the top-level
    `uprightmpc2py` names. No GPU call here: classes and signatures only (constructing a controller needs the device)."""
    import subprocess
    import sys
    from robobee3d_amd import _lib
    assert os.path.exists(_lib.ext_path()), "run __graft_entry__.build()"
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import uprightmpc2py as m\n"
            "from robobee3d_amd import uprightmpc2py as w, _uprightmpc2py as n\n"
            "assert w.binding() == 'pybind11' and m.UprightMPC2C is n.UprightMPC2C and m.WLCon is n.WLCon\n"
            "d = n.UprightMPC2C.update.__doc__\n"
            "assert all(k in d for k in ('p0', 'R0', 'dq0', 'pdes', 'dpdes', 'sdes', 'actualT0')) and '= -1.0' in d\n"
            "assert all(hasattr(n.UprightMPC2C, k) for k in ('update', 'vectors', 'matrices', 'status', 'set_compat'))\n"
            "assert hasattr(n.WLCon, 'update')\n"
            "print('ok')\n") % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "UMPC_LIB", "UMPC_PY_BINDING")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/", env=env)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]
    # ... and the ctypes binding stays selectable
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\nfrom robobee3d_amd import uprightmpc2py as w\n"
                        "assert w.binding() == 'ctypes' and w.UprightMPC2C is w.UprightMPC2C_ctypes\nprint('ok')" % ROOT],
                       capture_output=True, text=True, cwd="/", env=dict(env, UMPC_PY_BINDING="ctypes"))
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]
    with open(os.path.join(ROOT, "robobee3d_amd", "csrc", "uprightmpc2py_ext.cpp")) as f:
        src = f.read()
    assert "umpcInit(" in src and "umpcUpdate(" in src and "wlConInit(" in src and "wlConUpdate(" in src

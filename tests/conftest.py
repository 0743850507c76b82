import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def structure():
    return golden("structure.npz")


@pytest.fixture(scope="session")
def oracle_built():
    import oraclebind
    oraclebind.build()
    return oraclebind


# ---------------------------------------------------------------------------
# Achieved parity margins: every parity test records its measured worst case here and the table is printed at
# the end of the run (also under -q), so the log of a green run carries the numbers, not only "passed".
# ---------------------------------------------------------------------------
MARGINS = []


def record_margin(test, quantity, achieved, bound, note=""):
    MARGINS.append((test, quantity, float(achieved), float(bound), note))


@pytest.fixture
def margin(request):
    name = request.node.name

    def rec(quantity, achieved, bound, note=""):
        record_margin(name, quantity, achieved, bound, note)
        assert achieved <= bound, (name, quantity, achieved, bound)
    return rec


def pytest_terminal_summary(terminalreporter):
    if not MARGINS:
        return
    tr = terminalreporter
    tr.section("parity margins (achieved worst case vs asserted bound)")
    for test, q, a, b, note in MARGINS:
        tr.write_line("%-62s %-34s achieved %.3e  bound %.3e  (%.0f%%)%s"
                      % (test[:62], q[:34], a, b, 100.0 * a / b if b else 0.0, ("  " + note) if note else ""))
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        import json
        with open(os.path.join(out, "parity_margins.json"), "w") as f:
            json.dump([dict(test=t, quantity=q, achieved=a, bound=b, note=n) for t, q, a, b, n in MARGINS], f, indent=1)
    except OSError:
        pass

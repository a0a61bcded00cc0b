import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def pytest_sessionfinish(session, exitstatus):
    """Write the parity numbers measured in this session next to their bars (tests/util.check)."""
    try:
        from tests import util
    except Exception:
        return
    if not util.PARITY_LOG:
        return
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        import json
        rows = {}
        for k, vs in sorted(util.PARITY_LOG.items()):
            worst = max(vs, key=lambda v: v[0] / v[1] if v[1] > 0 else v[0])       # closest to (or furthest over) its bar
            rows[k] = {"measured_max": max(v[0] for v in vs), "worst_vs_bar": worst[0], "bar": worst[1], "n": len(vs),
                       "worst_case": worst[2], "over": [[v[2], v[0], v[1]] for v in vs if v[0] > v[1]][:40]}
        with open(os.path.join(out_dir, "parity_measured.json"), "w") as f:
            json.dump(rows, f, indent=1)
    except OSError:
        pass

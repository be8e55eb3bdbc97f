import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_json(npz, key):
    return json.loads(str(npz[key]))


def rel_l2(a, b):
    import torch
    a = torch.as_tensor(a, dtype=torch.float64).flatten()
    b = torch.as_tensor(b, dtype=torch.float64).flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# ---- worst-case error ledger: every parity check notes its error here; the summary is printed at the end of the run, so the
# GPU test log carries the margins (not only pass / fail) --------------------------------------------------------------
_WORST = {}


def note(family, err, detail=""):
    cur = _WORST.get(family)
    if cur is None or err > cur[0]:
        _WORST[family] = (float(err), str(detail), (cur[2] if cur else 0) + 1)
    else:
        _WORST[family] = (cur[0], cur[1], cur[2] + 1)


def check(family, got, want, tol, detail=""):
    """rel-L2(got, want) < tol, recorded under `family`."""
    e = rel_l2(got, want)
    note(family, e, detail)
    assert e < tol, (family, detail, e, tol)
    return e


def pytest_terminal_summary(terminalreporter):
    if not _WORST:
        return
    terminalreporter.write_sep("-", "worst relative-L2 error per parity family (gate in the test)")
    for fam in sorted(_WORST):
        e, d, n = _WORST[fam]
        terminalreporter.write_line(f"{fam:58s} {e:9.2e}   over {n:4d} checks   worst at: {d}")

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the loader's decoded-frame cache defaults to ~/.cache/mmego_amd: keep test runs inside a throw-away directory
if "MMEGO_CACHE_DIR" not in os.environ:
    import tempfile
    os.environ["MMEGO_CACHE_DIR"] = tempfile.mkdtemp(prefix="mmego_cache_")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(autouse=True)
def _stall_probe(request):
    """MMEGO_STALL_PROBE=<seconds>[:<logfile>]: native backtraces of every thread when a test runs longer (tests/stall_probe.py)."""
    from stall_probe import arm
    w = arm(request.node.nodeid)
    yield
    if w is not None:
        w.stop()


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def load_weights(module, npz, prefix=""):
    """Load an npz of state_dict arrays (keys optionally prefixed) into a module."""
    sd = {k[len(prefix):]: torch.tensor(npz[k]) for k in npz.files if k.startswith(prefix)}
    module.load_state_dict(sd)
    return module


def set_lstm_dropout(module, p):
    for m in module.modules():
        if isinstance(m, torch.nn.LSTM):
            m.dropout = p
        if hasattr(m, "lstm_dropout"):
            m.lstm_dropout = p


def check_pinned(npz, prefix, named_tensors, rtol, atol, bad_frac=0.0, hard_atol=None):
    """Compare tensors against full or '#s'-summarised pins written by make_golden.pin().

    ``bad_frac``: fraction of elements (over all tensors) allowed outside rtol/atol, provided they stay
    inside ``hard_atol`` (used for Adam-updated parameters, where a near-zero gradient whose sign is
    rounding noise moves a weight by +-lr in either implementation)."""
    worst = 0.0
    n_bad = n_all = 0
    for name, t in named_tensors:
        key = prefix + name
        a = t.detach().double().cpu().flatten()
        if key in npz.files:
            ref = torch.tensor(npz[key]).double().flatten()
            got = a
        elif key + "#s" in npz.files:
            s = npz[key + "#s"]
            stride = int(s[2])
            ref = torch.tensor(s[3:])
            got = a[::stride][:ref.numel()]
        else:
            raise KeyError(key)
        err = (got - ref).abs()
        tol = atol + rtol * ref.abs()
        bad = err > tol
        n_bad += int(bad.sum())
        n_all += bad.numel()
        if bad_frac == 0.0:
            assert not bad.any(), "%s: max err %.3e (tol %.3e) at %d/%d" % (
                key, err.max().item(), tol[err.argmax()].item(), int(bad.sum()), bad.numel())
        else:
            assert err.max().item() <= hard_atol, "%s: max err %.3e beyond hard bound %.1e" % (key, err.max().item(), hard_atol)
        worst = max(worst, (err / tol).max().item())
    assert n_bad <= bad_frac * n_all, "%d of %d elements outside tolerance (allowed %.3f)" % (n_bad, n_all, bad_frac)
    return worst


@pytest.fixture(scope="session")
def real16():
    return golden("real16.npz")

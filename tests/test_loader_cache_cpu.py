"""CPU: the packed decoded-frame cache of mmego_amd.data gives the same arrays and consumes numpy's RNG
exactly like a cold read, on a tiny synthetic Sample_data tree with the reference's .mat keys."""
import os

import numpy as np

from test_cli_gpu import _make_dataset


def test_cache_hit_equals_cold_read(tmp_path, monkeypatch):
    from mmego_amd import data
    root = str(tmp_path / "Sample_data")
    _make_dataset(root, np.random.default_rng(3))
    cache_dir = str(tmp_path / "cache")
    monkeypatch.setenv("MMEGO_CACHE_DIR", "off")
    np.random.seed(11)
    cold = data.PosePC(train=True, batch_length=8, root=root)
    state_cold = np.random.get_state()[1].copy()
    monkeypatch.setenv("MMEGO_CACHE_DIR", cache_dir)
    np.random.seed(11)
    first = data.PosePC(train=True, batch_length=8, root=root)          # writes the cache
    assert len([f for f in os.listdir(cache_dir) if f.endswith(".npz")]) == 1
    calls = []
    real = data.scio.loadmat
    monkeypatch.setattr(data.scio, "loadmat", lambda *a, **k: calls.append(a) or real(*a, **k))
    np.random.seed(11)
    warm = data.PosePC(train=True, batch_length=8, root=root)           # must not open a single .mat
    assert calls == []
    state_warm = np.random.get_state()[1].copy()
    assert np.array_equal(state_cold, state_warm)
    assert len(cold) == len(first) == len(warm) > 0
    for a, b, c in zip(cold._items, first._items, warm._items):
        assert a.dtype == b.dtype == c.dtype and a.shape == b.shape == c.shape
        assert np.array_equal(a, b) and np.array_equal(a, c)
    # touching a frame file invalidates the cache key
    some = os.path.join(root, "01", "s2", "frame_3.mat")
    os.utime(some, ns=(1, 1))
    np.random.seed(11)
    data.PosePC(train=True, batch_length=8, root=root)
    assert len(calls) > 0 and len([f for f in os.listdir(cache_dir) if f.endswith(".npz")]) == 2


def test_batch_indices_match_batches():
    from mmego_amd import data

    class D:
        _items = [np.arange(23)]

        def __len__(self):
            return 23
    a = [b[0] for b in data.batches(D(), 5, True, np.random.RandomState(4))]
    b = list(data.batch_indices(23, 5, True, np.random.RandomState(4)))
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and len(a) == len(b) == 5

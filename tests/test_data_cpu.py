"""CPU: the Sample_data loader reproduces the reference loader's arrays (needs the reference's data directory,
which exists only in the build container; skipped elsewhere)."""
import os

import numpy as np
import pytest

from conftest import golden

DATA = os.environ.get("MMEGO_DATA_ROOT", "/root/reference/Resource/Sample_data")


@pytest.mark.skipif(not os.path.isdir(DATA), reason="Sample_data not available")
def test_loader_matches_reference_arrays():
    from mmego_amd.config import Config
    from mmego_amd.data import PosePC, batches
    Config.data_root = DATA
    real = golden("real16.npz")
    np.random.seed(0)                                    # the seed make_golden.py used before the reference loader
    ds = PosePC(train=False, vis=True, batch_length=20)
    assert ds.data_ti_.shape == (835, 20, 128, 6) and ds.data_ti_.dtype == np.float32
    assert ds.data_key_.shape == (835, 20, 21, 3) and ds.imu_.shape == (835, 20, 20, 15) and ds.skl_.shape == (835, 20, 3)
    sel = real["sel"]
    assert np.array_equal(ds.data_ti_[sel], real["x"]), "padded point clouds must be bit-identical under the same seed"
    assert np.array_equal(ds.data_key_[sel].astype(np.float32), real["target"])
    assert np.array_equal(ds.skl_[sel].astype(np.float32), real["skl"])
    assert np.array_equal(ds.R_R0R_[sel].astype(np.float32), real["R"])
    assert np.array_equal(ds.imu_[sel].astype(np.float32), real["imu"])
    assert len(ds) == 835 and len(ds[0]) == 9
    b = next(batches(ds, 4, False))
    assert b[0].shape == (4, 20, 128, 6) and len(b) == 9
    # train/test split sizes of the seeded shuffle (80/20)
    np.random.seed(0)
    tr = PosePC(train=True, batch_length=20)
    assert len(tr) == int(835 * 0.8) and len(tr[0]) == 8

"""Drop-in import path of the reference (Util/Universal_Util/Dataset_sample.py:12)."""
from mmego_amd.data import PosePC  # noqa: F401

"""Drop-in import path of the reference (Util/Universal_Util/Utils.py:14,274,284)."""
from mmego_amd.utils import EarlyStopping, Transform2H, Transform2R, angle_minus  # noqa: F401

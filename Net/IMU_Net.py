"""Drop-in import path of the reference (Net/IMU_Net.py:50): `from Net.IMU_Net import IMUNet`."""
from mmego_amd.nets import IMUNet  # noqa: F401

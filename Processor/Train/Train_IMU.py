"""Drop-in import path of the reference (Processor/Train/Train_IMU.py:38)."""
from mmego_amd.processors import ImuTrainer as MMEgo  # noqa: F401

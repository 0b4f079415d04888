"""Drop-in import path of the reference (Processor/Train/Train_Upper.py:20)."""
from mmego_amd.processors import UpperTrainer as MMEgo  # noqa: F401

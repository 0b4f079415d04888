"""Drop-in import path of the reference (Processor/Train/Train_Lower.py:23)."""
from mmego_amd.processors import LowerTrainer as MMEgo  # noqa: F401

"""Drop-in import path of the reference (Net/Lower_Net.py:170)."""
from mmego_amd.nets import LowerNet  # noqa: F401

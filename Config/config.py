"""Drop-in import path of the reference (Config/config.py:11)."""
from mmego_amd.config import Config  # noqa: F401

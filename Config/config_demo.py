"""Drop-in import path of the reference (Config/config_demo.py:11)."""
from mmego_amd.config import ConfigDemo as Config  # noqa: F401

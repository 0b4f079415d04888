"""Drop-in import path of the reference (Net/Upper_Net.py:367,406)."""
from mmego_amd.nets import UpperNet  # noqa: F401
from mmego_amd.nets_local import UpperNetwlocal  # noqa: F401

"""Drop-in import path of the reference (Net/GCN.py:281): the ST-GCN parameter container."""
from mmego_amd.nets import Model  # noqa: F401

"""Drop-in import path of the reference (Processor/Test/Demo_test.py:23)."""
from mmego_amd.processors import Evaluator as MMEgo  # noqa: F401

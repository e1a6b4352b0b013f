"""`modules.iterative` drop-in (/root/reference/modules/iterative.py:15-74)."""
from mpcore.iterative import TensorTransform, iterative_loss, sort_channels_descending_norm  # noqa: F401

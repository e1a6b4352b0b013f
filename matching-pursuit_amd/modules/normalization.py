"""`modules.normalization.unit_norm` drop-in (/root/reference/modules/normalization.py:4-6)."""
from mpcore.matchingpursuit import unit_norm  # noqa: F401

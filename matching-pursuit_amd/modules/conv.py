"""`modules.conv` drop-in (/root/reference/modules/conv.py:4-53)."""
from mpcore.matchingpursuit import fft_convolve, torch_conv  # noqa: F401

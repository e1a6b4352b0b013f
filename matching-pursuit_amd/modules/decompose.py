"""`modules.decompose` drop-in (/root/reference/modules/decompose.py)."""
from mpcore.decompose import fft_frequency_decompose, fft_frequency_recompose, fft_resample  # noqa: F401

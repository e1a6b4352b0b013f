"""Octave band split / resample / recompose used by the multiband wrapper
(/root/reference/modules/decompose.py:5-82).  Pure spectral slicing on `torch.fft` (rocFFT on the
device); it runs once per batch on either side of the per-band matching pursuit.
"""
import torch


def fft_frequency_decompose(x: torch.Tensor, min_size: int):
    """x [B, C, T] -> {size: band [B, C, size]} for size = min_size, 2 min_size, ... <= T
    (decompose.py:5-33).  The lowest band keeps bins [0, min_size/2]; band `size` keeps bins
    [size/4, size/2] of the spectrum and is synthesised at `size` samples."""
    spec = torch.fft.rfft(x, norm="ortho")
    bands = {}
    size = min_size
    while size <= x.shape[-1]:
        part = spec[..., : size // 2 + 1]
        if size > min_size:
            keep = torch.zeros(part.shape[-1], device=x.device)
            keep[size // 4: size // 2 + 1] = 1
            part = part * keep
        bands[size] = torch.fft.irfft(part, n=size, norm="ortho")
        size *= 2
    return bands


def fft_resample(x: torch.Tensor, desired_size: int, is_lowest_band: bool):
    """x [B, C, T] -> [B, C, desired_size] by zero-extending the spectrum (decompose.py:36-73): the lowest
    band is copied whole, every other band contributes only its upper half (the part it owns)."""
    batch, channels, _ = x.shape
    spec = torch.fft.rfft(x, norm="ortho")
    n = spec.shape[-1]
    out = torch.zeros(batch, channels, desired_size // 2 + 1, dtype=torch.complex64, device=x.device)
    if is_lowest_band:
        out[..., :n] = spec
    else:
        out[..., n // 2: n] = spec[..., n // 2:]
    return torch.fft.irfft(out, n=desired_size, norm="ortho")


def fft_frequency_recompose(d, desired_size: int):
    """{size: band} -> their sum at `desired_size` samples (decompose.py:76-82)."""
    lowest = min(d.keys())
    total = 0
    for size, band in d.items():
        total = total + fft_resample(band, desired_size, size == lowest)
    return total

"""Mirror of the two live functions of the reference's `modules.iterative`
(/root/reference/modules/iterative.py:18-74).  `IterativeDecomposer` (:77-229) raises
NotImplementedError in its constructor in the reference and is therefore not mirrored.

The greedy residual-energy loss walks the (optionally loudness-sorted) event channels,
subtracting each from a running residual in the transform domain and scoring how much L1
energy each removes.  Written here over the whole [B, E, F] block at once: the running
residuals are `target - cumsum(channels)`, so all E+1 norms come from one cumulative sum
instead of E dependent passes.
"""
from typing import Callable

import torch

TensorTransform = Callable[[torch.Tensor], torch.Tensor]


def _sorted_by_l1(x: torch.Tensor) -> torch.Tensor:
    """Channels of x [B, E, F] reordered loudest (largest L1 norm) first."""
    order = torch.argsort(x.abs().sum(dim=-1), dim=-1, descending=True)
    return torch.take_along_dim(x, order[:, :, None], dim=1)


def sort_channels_descending_norm(x: torch.Tensor) -> torch.Tensor:
    """modules/iterative.py:18-22."""
    return _sorted_by_l1(x)


def iterative_loss(
        target_audio: torch.Tensor,
        recon_channels: torch.Tensor,
        transform: TensorTransform,
        return_residual: bool = False,
        ratio_loss: bool = False,
        sort_channels: bool = True):
    """modules/iterative.py:24-74.

    target_audio [B, 1, T], recon_channels [B, E, T]; `transform` maps [B, C, T] to any
    [B, C, ...] feature tensor.  Returns the scalar loss, or (final residual [B, F], loss).
    """
    batch, _, time = target_audio.shape
    batch_size, n_events, time = recon_channels.shape
    target = transform(target_audio.view(batch, 1, time)).reshape(batch, -1)          # :38-39
    channels = transform(recon_channels.view(batch, n_events, time)).reshape(batch, n_events, -1)  # :42-43
    if sort_channels:
        channels = _sorted_by_l1(channels)                                             # :47-51
    # residual after i channels, i = 0..E  (:58-63)
    running = target[:, None, :] - torch.cumsum(channels, dim=1)
    norms = torch.cat([target.abs().sum(-1, keepdim=True), running.abs().sum(-1)], dim=1)  # [B, E+1]
    start, end = norms[:, :-1], norms[:, 1:]
    if ratio_loss:
        loss = (end / (start + 1e-12)).sum()                                           # :64-65
    else:
        loss = (end - start).sum()                                                     # :66-68
    if n_events == 0:
        loss = loss + 0 * target.sum()
    residual = running[:, -1, :] if n_events > 0 else target
    if return_residual:
        return residual, loss
    return loss

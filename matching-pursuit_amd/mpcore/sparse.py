"""Mirror of the two helpers of the reference's `modules.sparse` that sit on the matching-pursuit
surface (/root/reference/modules/sparse.py:29-89): `soft_dirac` (used by sparse_feature_map,
matchingpursuit.py:100) and `sparsify2` (the top-k selector of the gradient-trained model, mp.py:61).
Plain torch ops, device-agnostic: both are thin index manipulations next to the correlation they follow.
"""
import torch


def soft_dirac(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """Forward: one-hot of the (softmax) argmax along `dim`; backward: the softmax's gradient
    (straight-through), modules/sparse.py:29-43."""
    soft = torch.softmax(x, dim=dim)
    index = torch.argmax(soft, dim=dim, keepdim=True)
    hard = torch.zeros_like(soft).scatter_(dim, index, 1.0)
    return soft + (hard - soft).detach()


def sparsify2(x: torch.Tensor, n_to_keep: int = 8):
    """Top-k over the flattened (channels x time) plane of x [B, C, T] (modules/sparse.py:46-89):
        sparse  [B, C, T]   x with everything but the k largest entries zeroed
        packed  [B, k, T]   row j holds the j-th largest value at its time index
        one_hot [B, k, C]   row j holds the j-th largest value at its channel index
    """
    batch, channels, time = x.shape
    flat = x.reshape(batch, -1)
    values, indices = torch.topk(flat, k=n_to_keep, dim=-1)
    ch = indices // time
    t = indices % time
    rows = torch.arange(n_to_keep, device=x.device)[None, :]
    sparse = torch.zeros_like(flat).scatter(-1, indices, values).view(batch, channels, time)
    packed = torch.zeros(batch, n_to_keep * time, device=x.device, dtype=x.dtype)
    packed = packed.scatter(-1, rows * time + t, values).view(batch, n_to_keep, time)
    one_hot = torch.zeros(batch, n_to_keep * channels, device=x.device, dtype=x.dtype)
    one_hot = one_hot.scatter(-1, rows * channels + ch, values).view(batch, n_to_keep, channels)
    return sparse, packed, one_hot

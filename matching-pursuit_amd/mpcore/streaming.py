"""Streaming / overlap-add encode of audio longer than one segment.

The reference has no matching-pursuit streaming of its own; the shape comes from its long-audio loops
(`iterativedecomposition.py:275-319`: windows of one segment at a 50 % hop, each window decomposed on the
RESIDUAL the previous windows left behind, the pieces overlap-added; `experiments/archive/e_2023_4_23/
experiment.py:24-73`: all windows of a recording encoded as one batch).  Here:

    encode_streaming(audio[B, T], d, window, hop, n_steps)  ->  StreamCode
        order="sequential":  window w is encoded on the residual left by windows 0..w-1 (the carried-residual
                             semantics of `Model.streaming`); one native encode of B segments per window.
        order="even_odd":    windows 0, 2, 4, ... of every recording as ONE batch, then 1, 3, 5, ... on what is
                             left (at a 50 % hop windows of equal parity never overlap): two native encodes of
                             B * W / 2 segments, the batch-everything shape of e_2023_4_23.
    decode_streaming(code, T)  ->  audio[B, T]        (one scatter launch; events carry global sample positions)

Every window is an ordinary `mp_encode_f32` segment: an atom that starts less than L samples before the end of
its window is cropped there, exactly as at the end of a segment (modules/matchingpursuit.py:20-58 crops at N);
what it leaves is picked up by the next window.  decode(code) + code.residual == audio up to fp32 rounding.
"""
from dataclasses import dataclass

import torch

from . import _native
from .matchingpursuit import _compute_device


@dataclass
class StreamCode:
    atom: torch.Tensor       # [B, W, K] int64
    position: torch.Tensor   # [B, W, K] int64, sample index in the recording where the atom starts
    gain: torch.Tensor       # [B, W, K] float32
    residual: torch.Tensor   # [B, T] float32
    dict_unit: torch.Tensor  # [A, L] float32
    window: int
    hop: int

    @property
    def n_windows(self):
        return self.atom.shape[1]


def n_windows(total, window, hop):
    """Windows needed to cover `total` samples: starts 0, hop, 2 hop, ... until a window reaches the end."""
    if total <= window:
        return 1
    return 1 + -(-(total - window) // hop)


def encode_streaming(audio, d, window, hop=None, n_steps=16, order="sequential", path=None):
    if audio.dim() == 3:
        audio = audio[:, 0, :]
    if audio.dim() != 2:
        raise ValueError("encode_streaming expects audio of shape (batch, samples) or (batch, 1, samples)")
    hop = window // 2 if hop is None else int(hop)
    if not 0 < hop <= window:
        raise ValueError("hop must be in (0, window]")
    if order not in ("sequential", "even_odd"):
        raise ValueError("order must be 'sequential' or 'even_odd'")
    if order == "even_odd" and 2 * hop < window:
        raise ValueError("order='even_odd' needs hop >= window / 2 (windows of equal parity must not overlap)")
    dev = _compute_device(audio)
    out_dev = audio.device
    B, T = audio.shape
    W = n_windows(T, window, hop)
    padded = (W - 1) * hop + window
    res = torch.zeros(B, padded, device=dev, dtype=torch.float32)
    res[:, :T] = audio.to(dev, torch.float32)
    du = _native.unit_norm(d.to(dev))
    K = int(n_steps)
    atom = torch.zeros(B, W, K, dtype=torch.int64, device=dev)
    pos = torch.zeros(B, W, K, dtype=torch.int64, device=dev)
    gain = torch.zeros(B, W, K, dtype=torch.float32, device=dev)

    def run(seg):
        if path is None:
            return _native.encode_checked(seg, du, K)
        return _native.encode(seg, du, K, path=path)

    if order == "sequential":
        for w in range(W):
            s = w * hop
            a, l, g, r = run(res[:, s:s + window].contiguous())
            res[:, s:s + window] = r
            atom[:, w], pos[:, w], gain[:, w] = a, l + s, g
    else:
        for parity in (0, 1):
            nw = len(range(parity, W, 2))
            if not nw:
                continue
            # windows parity, parity + 2, ... of a recording start 2 hop apart and never overlap (2 hop >= window): ONE strided
            # view [B, nw, window] over the residual -- gathered with one copy, written back with one copy, the events' global
            # positions with one broadcast add (the hand-over between the two batches used to be a Python loop over the
            # windows: a quarter of the call at 63 windows, bench.py: variants.streaming_even_odd)
            view = res.as_strided((B, nw, window), (res.stride(0), 2 * hop, 1), parity * hop)
            a, l, g, r = run(view.reshape(B * nw, window))
            view.copy_(r.view(B, nw, window))
            starts = (torch.arange(nw, device=dev, dtype=torch.int64) * (2 * hop) + parity * hop)[None, :, None]
            atom[:, parity::2] = a.view(B, nw, K)
            pos[:, parity::2] = l.view(B, nw, K) + starts
            gain[:, parity::2] = g.view(B, nw, K)
    return StreamCode(atom.to(out_dev), pos.to(out_dev), gain.to(out_dev), res[:, :T].contiguous().to(out_dev),
                      du, int(window), hop)


def decode_streaming(code, total=None):
    """Sum of all events at their global positions, [B, T].  An atom cropped at the end of its window is cropped
    here too: the window end is (window index) * hop + window."""
    dev = _compute_device(code.gain)
    B, W, K = code.atom.shape
    T = code.residual.shape[1] if total is None else int(total)
    L = code.dict_unit.shape[1]
    padded = (W - 1) * code.hop + code.window
    out = torch.zeros(B, padded, device=dev, dtype=torch.float32)
    if K == 0:
        return out[:, :T].to(code.gain.device)
    du = code.dict_unit.to(dev)
    a = code.atom.to(dev).reshape(B, -1)
    p = code.position.to(dev).reshape(B, -1)
    g = code.gain.to(dev).reshape(B, -1)
    ends = (torch.arange(W, device=dev) * code.hop + code.window)[None, :, None].expand(B, W, K).reshape(B, -1)
    j = torch.arange(L, device=dev)
    idx = p[:, :, None] + j[None, None, :]
    ok = idx < ends[:, :, None]
    vals = torch.where(ok, g[:, :, None] * du[a], torch.zeros((), device=dev))
    out.scatter_add_(1, idx.clamp(max=padded - 1).reshape(B, -1), vals.reshape(B, -1))
    return out[:, :T].to(code.gain.device)

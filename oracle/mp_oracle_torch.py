"""The reference's hot loop restated in torch CPU ops -- the second CPU baseline of SURVEY.md section 8(d).

TEST INFRASTRUCTURE ONLY (see the header of mp_oracle.c): importable from tests/ and from bench.py's
cpu_baseline leg; the product never imports it.

What the reference runs on a CPU is `F.conv1d` (oneDNN) + `torch.max` + a scatter per step
(/root/reference/modules/matchingpursuit.py:269-328).  oracle/mp_oracle.c restates that arithmetic in plain C (one
fp32 fma chain per map value, OpenMP over atoms) and is what parity is held against; THIS file restates it with the very
torch operators the reference calls, so that the timed CPU baseline also shows what those operators (oneDNN's
convolution, ATen's reduction) do on the host's cores.  Its picks equal the C oracle's wherever the top-2 gap exceeds
fp32 reordering noise (tests/test_oracle_golden.py holds it to the reference's fixtures and to the C oracle).
"""
import time

import numpy as np
import torch
from torch.nn import functional as F


def unit_norm(d, eps=1e-8):
    """modules/normalization.py:4-6."""
    d = torch.as_tensor(d, dtype=torch.float32)
    return d / (torch.norm(d, dim=-1, keepdim=True) + eps)


def encode(signal, dict_unit, n_steps):
    """sparse_code's loop, modules/matchingpursuit.py:269-328, on CPU tensors: per step
    fm = conv1d(pad(residual, (0, L)), d.view(A, 1, L))[..., :N] (:275-277), value, index = max over atom x lag
    (:298-303), residual[p : p + L] -= d[atom] * value, cropped at N (:305, :326-328 -- the scatter into a 3 N buffer
    and its crop are the same subtraction).  -> dict(atom [B, K] int64, lag [B, K] int64, gain [B, K] f32, residual [B, N])."""
    x = torch.as_tensor(signal, dtype=torch.float32)
    d = torch.as_tensor(dict_unit, dtype=torch.float32)
    B, N = x.shape
    A, L = d.shape
    residual = x.clone()
    w = d.view(A, 1, L)
    atoms, lags, gains = [], [], []
    rows = torch.arange(B)
    j = torch.arange(L)
    with torch.no_grad():
        for _ in range(int(n_steps)):
            fm = F.conv1d(F.pad(residual[:, None, :], (0, L)), w)[..., :N]          # :275-277
            value, mx = torch.max(fm.reshape(B, -1), dim=-1)                         # :298-299
            atom = mx // N                                                           # :302
            lag = mx % N                                                             # :303
            at = d[atom] * value[:, None]                                            # :305
            pos = lag[:, None] + j[None, :]
            ok = pos < N
            residual[rows[:, None].expand_as(pos)[ok], pos[ok]] -= at[ok]            # :326-328, cropped at N
            atoms.append(atom)
            lags.append(lag)
            gains.append(value)
    if not atoms:
        z = torch.zeros((B, 0))
        return dict(atom=z.long().numpy(), lag=z.long().numpy(), gain=z.float().numpy(), residual=residual.numpy())
    return dict(atom=torch.stack(atoms, 1).numpy(), lag=torch.stack(lags, 1).numpy(),
                gain=torch.stack(gains, 1).numpy(), residual=residual.numpy())


def timed_encode(signal, dict_unit, n_steps, threads):
    """-> (segment-iterations/s, seconds, result) with torch.set_num_threads(threads) for the duration."""
    before = torch.get_num_threads()
    torch.set_num_threads(int(threads))
    try:
        encode(signal[:1], dict_unit, 1)   # warm the thread pool and oneDNN's primitive cache
        t0 = time.perf_counter()
        out = encode(signal, dict_unit, n_steps)
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(before)
    return np.shape(signal)[0] * int(n_steps) / dt, dt, out

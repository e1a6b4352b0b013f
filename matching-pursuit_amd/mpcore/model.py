"""The reference's gradient-trained dictionary model (/root/reference/mp.py:32-107) on the native encoder.

`MatchingPursuit` keeps mp.py's constructor, parameter (`atoms` [1, A, L], raw, U(-0.01, 0.01)) and
`forward(audio[B,1,N]) -> channels[B,K,N]`.  What the reference does per step (mp.py:58-65):

    spec     = fft_convolve(residual, atoms_padded)        # CONVOLUTION, modules/transfer.py:548-569
    v, a, t  = top-1 of spec over atom x time              # sparsify2(spec, 1), modules/sparse.py:46-89
    b        = v * atoms[a]  (*)  v * delta(t)             # = v^2 * atoms[a] placed at t, cropped to N
    residual = residual - b ;  channels[:, i] = b

Here the K analysis steps (the whole cost: a feature map over A x N per step) run in one call of
`mp_encode_conv_f32`; the channels and the gradient are then rebuilt from the K events with window-sized
tensor operations (B x L per step).  The picks (a, t) are constants of the graph, exactly as argmax /
top-k indices are in the reference; the value v and the placed atom carry the gradient:

    v_i = sum_k r_i[t_i - k] * atoms[a_i][k],    b_i[n] = v_i^2 * atoms[a_i][n - t_i],    r_{i+1} = r_i - b_i.

`train_step` adds what config 5 of BASELINE.json asks for: the loss of the reference's training loop
(`iterative_loss`, mp.py:104) and one all-reduce of the [A, L] dictionary gradient across ranks.
"""
import torch
from torch import nn

from . import _native, dist as _dist
from .iterative import iterative_loss
from .matchingpursuit import _compute_device


def _sum_rows_by_atom(rows, a_idx, n_atoms):
    """g[a] = sum of the event rows [B, K, L] whose atom is a.  As a product with the one-hot selection matrix
    (a small GEMM, deterministic) while that matrix is small; index_add_ (atomics -- 250 us at the config-5 shape,
    where many events share an atom) beyond."""
    B, K, L = rows.shape
    if n_atoms * B * K <= (1 << 24):
        onehot = torch.zeros(n_atoms, B * K, device=rows.device, dtype=rows.dtype)
        onehot.scatter_(0, a_idx.reshape(1, -1), 1.0)
        return onehot @ rows.reshape(B * K, L)
    return torch.zeros(n_atoms, L, device=rows.device, dtype=rows.dtype).index_add_(0, a_idx.reshape(-1),
                                                                                     rows.reshape(B * K, L))


class _ConvModelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, audio, atoms, n_iterations, path):
        # audio [B, N], atoms [A, L]
        B, N = audio.shape
        A, L = atoms.shape
        with torch.no_grad():
            a_idx, t_idx, v, residual = _native.encode(audio, atoms, n_iterations, path=path, conv_model=True)
            if path == _native.MP_PATH_FFT and bool(torch.isnan(v).any()):  # a screen overflowed: exact fallback
                a_idx, t_idx, v, residual = _native.encode(audio, atoms, n_iterations,
                                                           path=_native.MP_PATH_INCREMENTAL, conv_model=True)
            K = n_iterations
            j = torch.arange(L, device=audio.device)
            pos = t_idx[:, :, None] + j[None, None, :]                      # [B, K, L] output sample of atom tap j
            ok = pos < N
            vals = (v * v)[:, :, None] * atoms[a_idx]                       # v^2 * atom
            channels = torch.zeros(B, K, N, device=audio.device, dtype=audio.dtype)
            channels.scatter_add_(2, pos.clamp(max=N - 1), torch.where(ok, vals, torch.zeros_like(vals)))
        ctx.save_for_backward(atoms, a_idx, t_idx, v, residual)
        ctx.shape = (B, N, A, L, K)
        return channels

    @staticmethod
    def backward(ctx, grad_channels):
        atoms, a_idx, t_idx, v, r = ctx.saved_tensors
        B, N, A, L, K = ctx.shape
        # the whole reverse walk in one launch (mp_conv_model_backward_f32), then one index_add over the events
        lam, rows = _native.conv_model_backward(atoms, a_idx, t_idx, v, r, grad_channels)
        return lam, _sum_rows_by_atom(rows, a_idx, A), None, None

    @staticmethod
    def backward_stepwise(ctx, grad_channels):
        """The same walk as ~17 tensor operations per step: the statement of what the kernel computes, kept for
        the tests (tests/test_gpu_api.py compares the two)."""
        atoms, a_idx, t_idx, v, r = ctx.saved_tensors
        B, N, A, L, K = ctx.shape
        dev = atoms.device
        j = torch.arange(L, device=dev)
        bidx = torch.arange(B, device=dev)[:, None]
        g_atoms = torch.zeros_like(atoms)
        lam = torch.zeros(B, N, device=dev, dtype=atoms.dtype)   # d loss / d r_{i+1}
        r = r.clone()                                             # r_K, walked back to r_0
        for i in range(K - 1, -1, -1):
            a = a_idx[:, i]
            t = t_idx[:, i]
            vi = v[:, i]
            d = atoms[a]                                          # [B, L]
            pos = t[:, None] + j[None, :]                         # where b_i lives
            ok = pos < N
            posc = pos.clamp(max=N - 1)
            # r_i = r_{i+1} + b_i
            b_i = torch.where(ok, (vi * vi)[:, None] * d, torch.zeros_like(d))
            r.scatter_add_(1, posc, b_i)
            gb = grad_channels[:, i, :] - lam                     # d loss / d b_i
            gw = torch.where(ok, gb.gather(1, posc), torch.zeros_like(d))
            dv = 2.0 * vi * (gw * d).sum(-1)                      # through b_i = v^2 * atom
            # window of r_i the value looked at: r_i[t - k], zero for negative indices
            back = t[:, None] - j[None, :]
            okb = back >= 0
            backc = back.clamp(min=0)
            rw = torch.where(okb, r.gather(1, backc), torch.zeros_like(d))
            g_atoms.index_add_(0, a, (vi * vi)[:, None] * gw + dv[:, None] * rw)
            # d loss / d r_i = lam (through r_{i+1} = r_i - b_i) + dv * d v_i / d r_i
            lam = lam.scatter_add(1, backc, torch.where(okb, dv[:, None] * d, torch.zeros_like(d)))
        return lam, g_atoms, None, None


class _ConvModelEventsFn(torch.autograd.Function):
    """The same analysis loop, returning each event's own support only: windows [B, K, L] with
    windows[b, i, j] = channel i of segment b at sample t_i + j (zero beyond the segment), plus the positions.
    A loss that can work on the events (stft_iterative_loss) never forms the dense [B, K, N] channels."""

    @staticmethod
    def forward(ctx, audio, atoms, n_iterations, path):
        B, N = audio.shape
        A, L = atoms.shape
        with torch.no_grad():
            a_idx, t_idx, v, residual = _native.encode(audio, atoms, n_iterations, path=path, conv_model=True)
            if path == _native.MP_PATH_FFT and bool(torch.isnan(v).any()):
                a_idx, t_idx, v, residual = _native.encode(audio, atoms, n_iterations,
                                                           path=_native.MP_PATH_INCREMENTAL, conv_model=True)
            j = torch.arange(L, device=audio.device)
            ok = (t_idx[:, :, None] + j[None, None, :]) < N
            windows = torch.where(ok, (v * v)[:, :, None] * atoms[a_idx], torch.zeros((), device=audio.device))
        ctx.save_for_backward(atoms, a_idx, t_idx, v, residual)
        ctx.shape = (B, N, A, L, n_iterations)
        ctx.mark_non_differentiable(t_idx)
        return windows, t_idx

    @staticmethod
    def backward(ctx, grad_windows, _grad_t):
        atoms, a_idx, t_idx, v, r = ctx.saved_tensors
        B, N, A, L, K = ctx.shape
        lam, rows = _native.conv_model_backward(atoms, a_idx, t_idx, v, r, grad_windows, windowed=True)
        return lam, _sum_rows_by_atom(rows, a_idx, A), None, None


_stft_windows = {}


def reference_stft(x, ws=2048, step=256):
    """modules/stft.py:7-36 with pad=True, as mp.py:71-73 calls it: [B, C, T] -> [B, C, T // step, ws // 2 + 1]."""
    frames = x.shape[-1] // step
    x = torch.nn.functional.pad(x, (0, ws)).unfold(-1, ws, step)
    key = (ws, str(x.device))
    win = _stft_windows.get(key)
    if win is None:
        win = _stft_windows[key] = torch.hann_window(ws, device=x.device)
    x = x * win[None, None, :]
    return torch.abs(torch.fft.rfft(x, norm="ortho"))[:, :, :frames, :]


_loss_consts = {}


def _loss_constants(ws, slots, L, B, dev):
    """Tensors of the event-form loss that depend on shapes only (the Hann window, index ramps): built once per
    (window, slots, atom length, batch, device) -- rebuilt per call they were ten small launches of a host-bound step."""
    key = (ws, slots, L, B, str(dev))
    c = _loss_consts.get(key)
    if c is None:
        if len(_loss_consts) > 16:
            _loss_consts.clear()
        c = _loss_consts[key] = dict(hann=torch.hann_window(ws, device=dev), s=torch.arange(slots, device=dev),
                                     n=torch.arange(ws, device=dev), j=torch.arange(L, device=dev),
                                     b=torch.arange(B, device=dev))
    return c


class _FramePiecesFn(torch.autograd.Function):
    """pieces[b, k, s, n] = hann[n] * windows[b, k, (f0 + s) step + n - t]  (zero outside the event's window and beyond the
    last frame): the frames an event's support touches, windowed.  The backward GATHERS -- every window sample sits in
    at most ws / step + 1 frames, summed in a fixed order -- where autograd's backward of the forward's gather is a
    scatter-add of B K S ws atomics (5.7 M at the config-5 shape: 0.38 ms of a 2.7 ms train step, non-deterministic)."""

    @staticmethod
    def forward(ctx, windows, t_idx, f0, step, frames, c):
        B, K, L = windows.shape
        S, ws = c["s"].numel(), c["n"].numel()
        f = f0[:, :, None] + c["s"][None, None, :]                                          # [B, K, S]
        rel = f[..., None] * step + c["n"] - t_idx[:, :, None, None]                        # index into the window
        inside = (rel >= 0) & (rel < L) & (f[..., None] < frames)
        pieces = torch.where(inside, torch.gather(windows[:, :, None, :].expand(B, K, S, L), 3, rel.clamp(0, L - 1)),
                             torch.zeros((), device=windows.device)) * c["hann"]
        ctx.save_for_backward(t_idx, f0)
        ctx.meta = (step, frames, c, L)
        return pieces

    @staticmethod
    def backward(ctx, gp):
        t_idx, f0 = ctx.saved_tensors
        step, frames, c, L = ctx.meta
        B, K, S, ws = gp.shape
        f = f0[:, :, None] + c["s"][None, None, :]                                          # [B, K, S]
        # sample j of the window is entry n = j + t - f step of frame f
        n_idx = c["j"][None, None, :, None] + (t_idx[:, :, None] - f * step)[:, :, None, :]  # [B, K, L, S]
        ok = (n_idx >= 0) & (n_idx < ws) & (f < frames)[:, :, None, :]
        flat = c["s"][None, None, None, :] * ws + n_idx.clamp(0, ws - 1)
        vals = torch.gather((gp * c["hann"]).reshape(B, K, S * ws), 2, flat.reshape(B, K, L * S)).reshape(B, K, L, S)
        gw = torch.where(ok, vals, torch.zeros((), device=gp.device)).sum(-1)
        return gw, None, None, None, None, None


def stft_iterative_loss(model, target, ws=2048, step=256):
    """iterative_loss(target, model(target), transform) of the reference's training loop (mp.py:102-104) for ITS
    transform -- the magnitude STFT of modules/stft.py (window ws, hop step, pad=True) -- evaluated on the events.

    With ratio_loss=False the greedy loss telescopes (modules/iterative.py:58-68: sum_i (|res_i| - |res_{i-1}|)),
    so it equals  |T - sum_i S_i|_1 - |T|_1  with T = |STFT(target)| and S_i = |STFT(channel_i)|, whatever the
    order of the channels.  A channel is one atom at one position: S_i is zero outside the <= (ws + L) / step + 1
    frames its support touches, so only those frames are transformed (11 of 128 at the config-5 shape) and the
    dense [B, K, N] channels and their [B, K, frames, bins] spectrograms (270 MB there) are never formed.
    Same value and same gradients as the dense form (tests/test_gpu_api.py)."""
    batch, _, n = target.shape
    dev = _compute_device(model.atoms)
    path = model.path if model.path is not None else _native.default_path(model.atom_samples)
    x = target.reshape(batch, n).to(dev, torch.float32).contiguous()
    windows, t_idx = _ConvModelEventsFn.apply(x, model.atoms[0].to(dev), model.n_iterations, path)
    B, K, L = windows.shape
    frames = n // step
    slots = (ws + L) // step + 1
    c = _loss_constants(ws, slots, L, B, dev)
    f0 = torch.clamp(torch.div(t_idx - ws, step, rounding_mode="floor") + 1, min=0)       # first frame touched
    pieces = _FramePiecesFn.apply(windows, t_idx, f0, step, frames, c)                    # [B, K, S, ws], windowed
    mags = torch.abs(torch.fft.rfft(pieces, norm="ortho"))                                # [B, K, S, bins]
    bins = ws // 2 + 1
    f = f0[:, :, None] + c["s"][None, None, :]
    flat = (c["b"][:, None, None] * frames + f.clamp(max=frames - 1)).reshape(-1)
    summed = torch.zeros(B * frames, bins, device=dev).index_add_(0, flat, mags.reshape(-1, bins))
    t_spec = reference_stft(x[:, None, :], ws, step).reshape(B * frames, bins)
    return (t_spec - summed).abs().sum() - t_spec.abs().sum()


class MatchingPursuit(nn.Module):
    """mp.py:32-67.  `path`: the native schedule (default: the fastest exact one)."""

    def __init__(self, n_atoms: int, atom_samples: int, n_samples: int, n_iterations: int, path=None):
        super().__init__()
        self.n_atoms = n_atoms
        self.atom_samples = atom_samples
        self.n_samples = n_samples
        self.n_iterations = n_iterations
        self.path = path
        self.atoms = nn.Parameter(torch.zeros(1, n_atoms, atom_samples).uniform_(-0.01, 0.01))

    @property
    def normalized_atoms(self):
        """mp.py:43-49: the atoms zero-padded to the signal length (despite the name, not normalised)."""
        pad = torch.zeros(1, self.n_atoms, self.n_samples - self.atom_samples, device=self.atoms.device)
        return torch.cat([self.atoms, pad], dim=-1)

    def forward(self, audio: torch.Tensor) -> torch.Tensor:
        batch, _, time = audio.shape
        dev = _compute_device(self.atoms)
        path = self.path if self.path is not None else _native.default_path(self.atom_samples)
        x = audio.reshape(batch, time).to(dev, torch.float32).contiguous()
        channels = _ConvModelFn.apply(x, self.atoms[0].to(dev), self.n_iterations, path)
        return channels.to(audio.device)


def all_reduce_gradients(params, group=None, average=True):
    """One flat all-reduce of the gradients (RCCL over xGMI with the nccl backend); identity without a
    process group.  For the 512 x 512 dictionary that is a single 1 MiB message per step."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or not _dist.is_distributed(group):
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    flat = _dist.all_reduce_sum(flat, group)
    if average:
        flat = flat / _dist.rank_world(group)[1]
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def train_step(model, optimizer, target, transform, group=None):
    """One step of mp.py:96-107: forward, iterative_loss, backward, (all-reduce,) optimiser step.
    `target` is this rank's shard of the global batch.  Returns the local loss (a python float).
    transform: a callable, as iterative_loss takes it -- or ("stft", ws, step) for the reference's own transform
    (mp.py:71-73), which takes the event form of the loss (stft_iterative_loss)."""
    optimizer.zero_grad()
    if isinstance(transform, tuple) and transform and transform[0] == "stft":
        loss = stft_iterative_loss(model, target, *transform[1:])
    else:
        loss = iterative_loss(target, model(target), transform)
    loss.backward()
    all_reduce_gradients(list(model.parameters()), group)
    optimizer.step()
    return float(loss.item())

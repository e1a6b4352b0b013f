"""Host-side mirror of the reference's `modules.matchingpursuit` surface on top of libmpcore.

Same names, argument meaning, return structures and error behaviour as
/root/reference/modules/matchingpursuit.py (each function cites the lines it mirrors); the
arithmetic runs in hand-written HIP kernels through the C ABI (include/mpcore.h).  There is
no CPU implementation here: tensors that live on the CPU are moved to the current HIP
device, processed there and the results moved back; without a HIP device every entry point
raises `NativeError`.
"""
from collections import defaultdict

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import _native
from ._native import NativeError

__all__ = [
    "build_scatter_segments", "flatten_atom_dict", "sparse_code", "dictionary_learning_step",
    "sparse_feature_map", "sparse_coding_loss", "SparseCodingLoss", "sparse_code_to_differentiable_key_points",
    "unit_norm", "torch_conv",
    "fft_convolve", "EventList", "encode_packed", "first_selection_order", "group_events_by_atom",
]


# --------------------------------------------------------------------------------------------
# device plumbing
# --------------------------------------------------------------------------------------------
def _compute_device(t):
    if t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise NativeError(
            "mpcore needs a HIP device (MI355X): the input is on the CPU and no GPU is visible; "
            "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def unit_norm(x, dim=-1, epsilon=1e-8):
    """modules/normalization.py:4-6.  2-D fp32 along the last axis runs in mp_unit_norm_f32."""
    if x.dim() == 2 and dim in (-1, 1) and x.dtype == torch.float32 and not x.requires_grad:
        dev = _compute_device(x)
        return _native.unit_norm(x.to(dev), epsilon).to(x.device)
    n = torch.norm(x, dim=dim, keepdim=True)
    return x / (n + epsilon)


def torch_conv(signal, atom):
    """modules/conv.py:4-9: dense feature map [B, A, N] (mp_feature_map_f32)."""
    n_samples = signal.shape[-1]
    dev = _compute_device(signal)
    sig = signal.reshape(-1, n_samples).to(dev)
    fm = _native.feature_map(sig, atom.to(dev))
    return fm.to(signal.device)


def fft_convolve(signal, atoms, approx=None):
    """modules/conv.py:11-53.  The exact branch (approx is None, or not a slice / small int,
    :48-49) is the same map as torch_conv and is served by the direct kernel; the two
    approximate branches (band slice :24-29, top-k bins :30-47) need the spectra and are
    restated with torch.fft on the device."""
    batch = signal.shape[0]
    n_samples = signal.shape[-1]
    n_atoms, atom_size = atoms.shape
    is_slice = isinstance(approx, slice)
    is_topk = isinstance(approx, int) and not isinstance(approx, bool) and approx < n_samples
    if not is_slice and not is_topk:
        return torch_conv(signal, atoms).view(batch, n_atoms, n_samples)
    dev = _compute_device(signal)
    sig_t = F.pad(signal.to(dev), (0, atom_size))
    padded = F.pad(atoms.to(dev), (0, sig_t.shape[-1] - atom_size))
    sig = torch.fft.rfft(sig_t, dim=-1)
    atom = torch.fft.rfft(torch.flip(padded, dims=(-1,)), dim=-1)[None, ...]
    fm_spec = torch.zeros(batch, n_atoms, sig.shape[-1], device=dev, dtype=sig.dtype)
    if is_slice:
        fm_spec[..., approx] = sig[..., approx] * atom[..., approx]
    else:
        mags = torch.abs(sig)
        _, indices = torch.topk(mags, k=approx, dim=-1)
        sig_k = torch.gather(sig, dim=-1, index=indices)
        atom_k = torch.gather(atom.repeat(batch, 1, 1), dim=-1, index=indices.expand(batch, n_atoms, -1))
        fm_spec = torch.scatter(fm_spec, dim=-1, index=indices.expand(batch, n_atoms, -1), src=sig_k * atom_k)
    fm = torch.fft.irfft(fm_spec, dim=-1)
    fm = torch.roll(fm, 1, dims=(-1,))
    return fm[..., :n_samples].to(signal.device)


# --------------------------------------------------------------------------------------------
# events
# --------------------------------------------------------------------------------------------
class EventList(list):
    """A list of reference-style event tuples `(atom:int, batch:int, lag:LongTensor[1,1],
    scaled_atom:FloatTensor[1,1,L])` that also carries the packed device arrays it was built
    from, so the decoder can run as one kernel instead of a Python loop."""
    packed = None  # dict(atom, batch, lag, gain: 1-D device tensors, dict_unit: [A, L])


def _make_events(atom, lag, gain, dict_unit, out_device):
    """Packed [B,K] arrays -> (instances dict in first-selection order, flat selection-order
    EventList).  One host sync (atom.tolist())."""
    B, K = atom.shape
    L = dict_unit.shape[1]
    at_all = (dict_unit[atom] * gain[..., None]).to(out_device)  # d[atom] * value, :305
    lag_o = lag.to(out_device)
    atom_l = atom.tolist()
    # all B*K views in two calls (unbind walks the tensors in C++; indexing them one by one from Python costs
    # ~6 us per event, five times the encode itself at the headline shape)
    lag_v = lag_o.reshape(B * K, 1, 1).unbind(0)
    at_v = at_all.reshape(B * K, 1, 1, L).unbind(0)
    instances = defaultdict(list)
    for i in range(K):            # step-major, batch-minor: the order of :269 / :311
        for j in range(B):
            ai = atom_l[j][i]
            instances[ai].append((ai, j, lag_v[j * K + i], at_v[j * K + i]))
    return instances, atom_l


def encode_packed(signal, d, n_steps, path=None, flags=0):
    """The fast interface: signal [B,1,N] or [B,N], raw dictionary d [A,L] ->
    dict(atom[B,K], lag[B,K], gain[B,K], residual[B,N], dict_unit[A,L]) on the compute device, no
    Python objects per event.  path=None: the default schedule with its overflow check (one host
    sync); an explicit MP_PATH_* is fully asynchronous."""
    if signal.dim() == 3:
        signal = signal[:, 0, :]
    dev = _compute_device(signal)
    du = _native.unit_norm(d.to(dev))
    if path is None:
        atom, lag, gain, residual = _native.encode_checked(signal.to(dev), du, n_steps, flags=flags)
    else:
        atom, lag, gain, residual = _native.encode(signal.to(dev), du, n_steps, path=path, flags=flags)
    return dict(atom=atom, lag=lag, gain=gain, residual=residual, dict_unit=du)


# --------------------------------------------------------------------------------------------
# scatter_segments  (modules/matchingpursuit.py:20-58)
# --------------------------------------------------------------------------------------------
def build_scatter_segments(n_samples, atom_size):

    def scatter_segments(x, inst):
        if isinstance(x, tuple):
            shape = tuple(x)
            base = None
        else:
            shape = tuple(x.shape)
            base = x  # the reference concatenates x itself between the pads (:33-34): out = x + events
        if len(shape) != 3:
            raise ValueError("scatter_segments expects a (batch, channels, samples) tensor or shape")
        channels = shape[1] if base is None else 1  # :24-29
        inst = inst if isinstance(inst, list) else list(inst)

        packed = getattr(inst, "packed", None)
        needs_grad = any(torch.is_tensor(e[3]) and e[3].requires_grad for e in inst) or (
            base is not None and base.requires_grad)
        if channels != 1 or shape[1] != 1 or needs_grad:
            return _scatter_torch(shape, base, inst, n_samples, atom_size, channels)

        # where the result lives (like the reference: with x, or with the events) and where it is computed
        if base is not None:
            ref_t = base
        elif packed is not None:
            ref_t = torch.empty(0, device=packed["out_device"])
        elif len(inst):
            ref_t = inst[0][3]
        else:
            ref_t = torch.empty(0, device="cuda" if torch.cuda.is_available() else "cpu")
        out_dev = ref_t.device
        dev = _compute_device(ref_t)
        B = shape[0]
        if base is None:
            out = torch.zeros((B, n_samples), dtype=torch.float32, device=dev)
        else:
            out = base.detach().to(dev, torch.float32).reshape(B, n_samples).clone()
        if len(inst):
            if packed is not None:
                _native.scatter(packed["atom"], packed["batch"], packed["lag"], packed["gain"],
                                packed["dict_unit"], out)
            else:
                rows = torch.cat([e[3].reshape(1, atom_size) for e in inst], dim=0).to(dev)
                batch = torch.tensor([int(e[1]) for e in inst], dtype=torch.int64, device=dev)
                lag = torch.tensor([int(e[2]) for e in inst], dtype=torch.int64, device=dev)
                _native.scatter_rows(rows, batch, lag, out)
        return out.view(B, 1, n_samples).to(out_dev)

    return scatter_segments


def _scatter_torch(shape, base, inst, n_samples, atom_size, channels):
    """Differentiable / multi-channel form of :33-56 in plain torch ops (no kernel needed:
    these branches are used for small decoder graphs, never in the encode loop)."""
    if base is None:
        dev = inst[0][3].device if len(inst) else torch.device("cpu")
        x = torch.zeros(*shape, device=dev)
    else:
        x = base
    target = torch.cat([torch.zeros_like(x), x, torch.zeros_like(x)], dim=-1)
    counter = {}
    for ai, j, p, a in inst:
        p = int(p)
        start = n_samples + p
        end = start + atom_size
        ch = counter.get(j, 0)
        if channels == 1:
            target[j, :, start:end] = target[j, :, start:end] + a.view(-1, atom_size)
        else:
            target[j, ch, start:end] = a.view(-1, atom_size)
        counter[j] = ch + 1
    return target[..., n_samples:n_samples * 2]


def flatten_atom_dict(atom_dict):
    """modules/matchingpursuit.py:61-65: concatenation of the per-atom lists (grouped by atom
    in first-selection order, NOT selection order)."""
    all_instances = EventList()
    for k, v in atom_dict.items():
        all_instances.extend(v)
    return all_instances


# --------------------------------------------------------------------------------------------
# sparse_code  (modules/matchingpursuit.py:229-345)
# --------------------------------------------------------------------------------------------
def sparse_code(
        signal,
        d,
        n_steps=100,
        device=None,
        approx=None,
        flatten=False,
        extract_atom_embedding=None,
        visit_key_point=None,
        return_residual=False,
        local_contrast_norm=False,
        return_sparse_feature_map=False,
        compute_feature_map=None,
        fft_convolution=False):
    batch, channels, time = signal.shape  # ValueError for non-3-D input, like :244
    if channels != 1 or d.dim() != 2:
        raise NotImplementedError("mpcore.sparse_code supports mono signals and [A, L] dictionaries")
    signal = signal.view(signal.shape[0], channels, -1)
    n_samples = signal.shape[-1]
    n_atoms, atom_size = d.shape[0], d.shape[-1]
    out_dev = signal.device
    dev = _compute_device(signal)

    d_unit = _native.unit_norm(d.to(dev))  # :254
    scatter_segments = build_scatter_segments(n_samples, atom_size)  # :259

    approximate = isinstance(approx, slice) or (
        isinstance(approx, int) and not isinstance(approx, bool) and approx < n_samples)
    dense = (compute_feature_map is not None or extract_atom_embedding is not None or
             visit_key_point is not None or approximate)

    if local_contrast_norm and not dense:  # :284-294 natively (mp_encode_lcn_f32)
        atom, lag, gain, residual = _native.encode_lcn(signal.to(dev)[:, 0, :], d_unit, n_steps)
        embeddings = None
    elif dense:
        atom, lag, gain, residual, embeddings = _sparse_code_dense(
            signal.to(dev), d_unit, n_steps, approx if approximate else None, extract_atom_embedding,
            visit_key_point, local_contrast_norm, compute_feature_map)
    else:
        atom, lag, gain, residual = _native.encode_checked(signal.to(dev)[:, 0, :], d_unit, n_steps)
        embeddings = None

    if extract_atom_embedding is not None:  # :332-333
        return embeddings, residual.view(batch, 1, n_samples).to(out_dev)

    instances, atom_l = _make_events(atom, lag, gain, d_unit, out_dev)

    if not flatten:  # :335-336
        return instances, scatter_segments

    flattened = flatten_atom_dict(instances)
    # packed arrays in the flattened (grouped-by-atom) order, so scatter(shape, events) is one kernel
    if len(flattened):
        K = atom.shape[1]
        perm, _ = group_events_by_atom(atom.cpu(), list(instances.keys()), n_atoms)
        idx = perm.to(dev)
        flattened.packed = dict(atom=atom.reshape(-1)[idx], batch=idx // K, lag=lag.reshape(-1)[idx],
                                gain=gain.reshape(-1)[idx], dict_unit=d_unit, out_device=out_dev)

    if return_residual:  # :337-339
        return flattened, scatter_segments, residual.view(batch, 1, n_samples).to(out_dev)
    if return_sparse_feature_map:  # :340-342, sfm[j, ai, p] += value (:317-318)
        sfm = torch.zeros(batch, n_atoms, n_samples, device=dev)
        bidx = torch.arange(batch, device=dev)[:, None].expand_as(atom)
        sfm.index_put_((bidx.reshape(-1), atom.reshape(-1), lag.reshape(-1)), gain.reshape(-1), accumulate=True)
        return flattened, scatter_segments, sfm.to(out_dev)
    return flattened, scatter_segments  # :343-345


def _sparse_code_dense(signal, d_unit, n_steps, approx, extract_atom_embedding, visit_key_point,
                       local_contrast_norm, compute_feature_map):
    """The hook-serving loop (:269-328): the dense map is materialised every step by
    mp_feature_map_f32 (or by the caller's compute_feature_map), so this costs A*N*4 bytes per
    segment per step -- correctness over speed, exactly as the reference does it."""
    B, _, N = signal.shape
    A, L = d_unit.shape
    dev = signal.device
    residual = signal[:, 0, :].to(torch.float32).clone()
    atoms, lags, gains, embeddings = [], [], [], []
    for _ in range(n_steps):
        if compute_feature_map is not None:
            fm = compute_feature_map(residual.view(B, 1, N), d_unit)  # :272-273
        elif approx is None:
            fm = _native.feature_map(residual, d_unit)  # :275-277
        else:
            fm = fft_convolve(residual.view(B, 1, N), d_unit, approx=approx)  # :280
        if extract_atom_embedding is not None:
            embeddings.append(extract_atom_embedding(fm, d_unit))  # :282-283
        if local_contrast_norm:  # :286-296
            feature_map = fm.view(B, 1, A, N)
            averages = F.avg_pool2d(feature_map, (9, 9), (1, 1), (4, 4))
            feature_map = (feature_map - averages).reshape(B, -1)
            fm = fm.reshape(B, -1)
            _, mx = torch.max(feature_map, dim=-1, keepdim=True)
            value = torch.gather(fm, dim=-1, index=mx)
        else:
            fm = fm.reshape(B, -1)
            value, mx = torch.max(fm, dim=-1, keepdim=True)  # :298-299
        atom_index = mx // N  # :302
        position = mx % N  # :303
        if visit_key_point is not None:  # :323-324
            at = d_unit[atom_index[:, 0]] * value
            for j in range(B):
                visit_key_point(fm[j].view(A, N), int(atom_index[j]), position[j].view(1), at[j].view(L))
        # residual -= scatter(d[atom] * value), cropped at N  (:305, :326-328)
        _native.scatter(atom_index[:, 0], torch.arange(B, device=dev), position[:, 0], -value[:, 0], d_unit,
                        residual)
        atoms.append(atom_index)
        lags.append(position)
        gains.append(value)
    if n_steps == 0:
        z = torch.zeros((B, 0), device=dev)
        return z.long(), z.long(), z.float(), residual, embeddings
    return torch.cat(atoms, 1), torch.cat(lags, 1), torch.cat(gains, 1).float(), residual, embeddings


def first_selection_order(atom):
    """atom [B, K] (selection order per segment) -> atoms in the order the reference's `instances`
    dict first sees them: steps outer, batch inner (:269, :311, :321)."""
    order, seen = [], set()
    for a in atom.t().reshape(-1).tolist():
        if a not in seen:
            seen.add(a)
            order.append(a)
    return order


def group_events_by_atom(atom, order, n_atoms):
    """Permutation of the flat [B*K] event index (b*K + k) that lists events grouped by atom in
    `order`, each group in (step, batch) order -- the layout of flatten_atom_dict (:61-65) -- and the
    number of events per group.  Atoms of `order` absent from `atom` (another rank's) get 0."""
    B, K = atom.shape
    a = atom.cpu().numpy()  # (numpy: a dozen small host tensor operations cost 2 ms here, these 0.3 ms)
    rank_of = np.full(n_atoms, len(order), dtype=np.int64)
    if len(order):
        rank_of[np.asarray(order, dtype=np.int64)] = np.arange(len(order), dtype=np.int64)
    r = rank_of[a]
    sort_key = (r * K + np.arange(K, dtype=np.int64)[None, :]) * max(B, 1) + np.arange(B, dtype=np.int64)[:, None]
    perm = np.argsort(sort_key.reshape(-1), kind="stable")
    counts = np.bincount(r.reshape(-1), minlength=len(order) + 1)[: len(order)].tolist()
    return torch.from_numpy(perm), counts


# --------------------------------------------------------------------------------------------
# dictionary_learning_step  (modules/matchingpursuit.py:348-419)
# --------------------------------------------------------------------------------------------
def dictionary_learning_step(
        signal,
        d,
        n_steps=100,
        device=None,
        approx=None,
        local_constrast_norm=False,
        compute_feature_map=None,
        fft_convolution=False,
        process_group=None):
    """Returns the updated, unit-normed dictionary; `d` itself is not modified (:365).

    `process_group` (extension): when given, `signal` is this rank's shard of a larger batch
    split contiguously over the group's ranks; the per-atom window sums (the only cross-segment
    quantity, :400-401) are all-reduced, and every rank returns the same dictionary as a
    single-device run over the concatenated batch.
    """
    batch, channels, time = signal.shape
    if channels != 1 or d.dim() != 2:
        raise NotImplementedError("mpcore.dictionary_learning_step supports mono signals and [A, L] dictionaries")
    n_samples = time
    n_atoms, atom_size = d.shape
    out_dev = d.device
    dev = _compute_device(signal)
    sig = signal.detach().to(dev, torch.float32).reshape(batch, n_samples)

    d_work = _native.unit_norm(d.detach().to(dev))  # :365 (a new tensor)
    residual = sig.clone()  # :367 -- the ORIGINAL signal, as in the reference

    dense = compute_feature_map is not None or isinstance(approx, slice) or (
        isinstance(approx, int) and not isinstance(approx, bool) and approx < n_samples)
    if local_constrast_norm and not dense:
        atom, lag, gain, _ = _native.encode_lcn(sig, d_work, n_steps, want_residual=False)
    elif dense:
        atom, lag, gain, _, _ = _sparse_code_dense(sig.view(batch, 1, n_samples), d_work, n_steps,
                                                   approx if not (approx is None) else None, None, None,
                                                   local_constrast_norm, compute_feature_map)
    else:
        atom, lag, gain, _ = _native.encode_checked(sig, d_work, n_steps, want_residual=False)

    K = atom.shape[1]
    # per-event payloads as materialised at encode time: a = d[atom] * value (:305), ||a|| (:410)
    rows = d_work[atom] * gain[..., None]  # [B, K, L]
    anorm = torch.norm(rows, dim=-1)  # [B, K]

    # global (step-major, batch-minor) first-selection order of atoms (:391, dict insertion order)
    from . import dist as _dist
    atom_global, _ = _dist.gather_batch(atom, process_group)
    order = first_selection_order(atom_global.cpu())
    perm, counts = group_events_by_atom(atom.cpu(), order, n_atoms)
    perm_d = perm.to(dev)
    ev_batch = (perm_d // K)
    ev_lag = lag.reshape(-1)[perm_d]
    ev_rows = rows.reshape(-1, atom_size)[perm_d]
    ev_norm = anorm.reshape(-1)[perm_d]

    if not _dist.is_distributed(process_group):
        # single device: the whole atom-by-atom loop in one launch (mp_dictionary_update_f32)
        counts_t = torch.as_tensor(counts, dtype=torch.int64)
        offsets = torch.zeros(len(order) + 1, dtype=torch.int64)
        offsets[1:] = torch.cumsum(counts_t, 0)
        _native.dictionary_update(residual, d_work, torch.as_tensor(order, dtype=torch.int64), offsets, ev_batch,
                                  ev_lag, ev_rows.contiguous(), ev_norm.contiguous())
        return _native.unit_norm(d_work).to(out_dev)  # :417-419

    sparse = torch.empty_like(residual)
    start = 0
    for oi, index in enumerate(order):
        n = counts[oi]
        sl = slice(start, start + n)
        start += n
        if n:
            sparse.zero_()
            _native.scatter_rows(ev_rows[sl], ev_batch[sl], ev_lag[sl], sparse)  # :395
            residual += sparse  # :396
            acc = _native.gather_sum(residual, ev_batch[sl], ev_lag[sl], atom_size)  # :400-401
        else:
            acc = torch.zeros(atom_size, dtype=torch.float64, device=dev)
        acc = _dist.all_reduce_sum(acc, process_group)
        new_atom = _native.unit_norm(acc.to(torch.float32).view(1, atom_size))  # :403-404
        d_work[index] = new_atom[0]  # :406
        if n:
            sparse.zero_()
            _native.scatter_rows(new_atom * ev_norm[sl, None], ev_batch[sl], ev_lag[sl], sparse)  # :408-414
            residual -= sparse  # :415
    return _native.unit_norm(d_work).to(out_dev)  # :417-419


# --------------------------------------------------------------------------------------------
# sparse_feature_map / sparse_coding_loss  (modules/matchingpursuit.py:68-146, 422-463)
# --------------------------------------------------------------------------------------------
class _SparseFeatureMapFn(torch.autograd.Function):
    """sparse_feature_map with the reference's gradient (modules/matchingpursuit.py:100-120).

    Forward: the K events come from the native encoder; the dense map holds each step's value at its argmax.
    Backward: what autograd does to the reference's loop, step by step from the last to the first:

        fm      += hard_i * f_i,   hard_i = softmax(f_i) + (onehot_i - softmax(f_i)).detach()     (sparse.py:29-43)
        r_{i+1}  = r_i - v_i * d[a_i] at p_i (cropped at N),   v_i = f_i[a_i, p_i]                (:103-120)

    With G the gradient arriving at fm and lam the gradient w.r.t. r_{i+1}:
        w      = s * (G f_i - <G f_i, s>)                 (through the softmax, s = softmax(f_i) over A*N)
        w[a_i, p_i] += G[a_i, p_i] - <lam[p_i : p_i+L], d[a_i]>     (through hard's value 1 and through v_i)
        lam    = lam + conv_transpose(w, d)[:N]
    f_i is recomputed densely per step by the native kernel (mp_feature_map_f32) from r_i, which is
    replayed backwards from the final residual; softmax and the transposed convolution are tensor ops.
    This costs K dense passes -- as the reference's own forward does."""

    @staticmethod
    def forward(ctx, x, d_unit, n_steps):
        # x [B, N] on the compute device
        atom, lag, gain, residual = _native.encode_checked(x.detach(), d_unit, n_steps)
        B, N = x.shape
        A = d_unit.shape[0]
        fm = torch.zeros(B, A, N, device=x.device)
        if n_steps > 0:
            bidx = torch.arange(B, device=x.device)[:, None].expand_as(atom)
            fm.index_put_((bidx.reshape(-1), atom.reshape(-1), lag.reshape(-1)), gain.reshape(-1), accumulate=True)
        ctx.save_for_backward(d_unit, atom, lag, gain, residual)
        ctx.mark_non_differentiable(atom, lag)
        return fm, residual, atom, lag, gain.clone()

    @staticmethod
    def backward(ctx, g_fm, g_res, _ga, _gl, _gg):
        d_unit, atom, lag, gain, r = ctx.saved_tensors
        B, N = r.shape
        A, L = d_unit.shape
        K = atom.shape[1]
        dev = r.device
        j = torch.arange(L, device=dev)
        bidx = torch.arange(B, device=dev)
        lam = g_res.clone() if g_res is not None else torch.zeros(B, N, device=dev)
        g_fm = g_fm if g_fm is not None else torch.zeros(B, A, N, device=dev)
        r = r.clone()
        d_spec = torch.fft.rfft(d_unit, n=N + L)                       # [A, F], once
        for i in range(K - 1, -1, -1):
            a, p, v = atom[:, i], lag[:, i], gain[:, i]
            da = d_unit[a]                                             # [B, L]
            pos = p[:, None] + j[None, :]
            ok = pos < N
            posc = pos.clamp(max=N - 1)
            r.scatter_add_(1, posc, torch.where(ok, v[:, None] * da, torch.zeros_like(da)))   # r_i
            f = _native.feature_map(r, d_unit)                         # [B, A, N], exact chains
            # softmax over the A*N cells of each segment, as two-stage reductions (lags, then atoms): a row of
            # 16.8 M cells reduced by torch.softmax / sum(dim=(1, 2)) is one workgroup's work -- 10 ms a call
            m = f.amax(dim=2).amax(dim=1)
            s = torch.exp(f - m[:, None, None])
            s = s / s.sum(dim=2).sum(dim=1)[:, None, None]
            gf = g_fm * f
            c = (gf * s).sum(dim=2).sum(dim=1)[:, None, None]
            w = s * (gf - c)
            lam_win = torch.where(ok, lam.gather(1, posc), torch.zeros_like(da))
            w[bidx, a, p] += g_fm[bidx, a, p] - (lam_win * da).sum(-1)
            # lam += conv_transpose1d(w, d)[:N] = sum_a (w_a * d_a)[:N], as one spectral accumulation: 13x fewer
            # flops than the dense contraction (2 A L N per segment) and none of MIOpen's search on first use
            ws = torch.fft.rfft(w, n=N + L)                            # [B, A, F]
            lam = lam + torch.fft.irfft((ws * d_spec[None]).sum(1), n=N + L)[:, :N]
        return lam, None, None


def sparse_feature_map(signal, d, n_steps=100, device=None, approx=None, pooling=None,
                       return_residual=False):
    """modules/matchingpursuit.py:68-125: the dense [B, A, N] map holding, per step, the feature-map value at
    that step's argmax (soft_dirac's forward is the one-hot of the argmax, :100-101).  Built from the encoder's
    events -- no softmax over A*N per step in the forward; differentiable w.r.t. the signal like the
    reference (see _SparseFeatureMapFn)."""
    signal = signal.view(signal.shape[0], 1, -1)
    batch, _, n_samples = signal.shape
    n_atoms, atom_size = d.shape
    out_dev = signal.device
    dev = _compute_device(signal)
    d_unit = _native.unit_norm(d.detach().to(dev))
    approximate = isinstance(approx, slice) or (
        isinstance(approx, int) and not isinstance(approx, bool) and approx < n_samples)
    if approximate:
        if signal.requires_grad:
            raise NotImplementedError(
                "mpcore.sparse_feature_map: gradients through the approximate (band-limited) correlation "
                "are not implemented")
        atom, lag, gain, residual, _ = _sparse_code_dense(signal.to(dev), d_unit, n_steps, approx, None, None,
                                                          False, None)
        fm = torch.zeros(batch, n_atoms, n_samples, device=dev)
        if n_steps > 0:
            bidx = torch.arange(batch, device=dev)[:, None].expand_as(atom)
            fm.index_put_((bidx.reshape(-1), atom.reshape(-1), lag.reshape(-1)), gain.reshape(-1), accumulate=True)
    else:
        x = signal.to(dev, torch.float32)[:, 0, :]
        fm, residual, _, _, _ = _SparseFeatureMapFn.apply(x, d_unit, n_steps)
    fm = fm.to(out_dev)
    if return_residual:
        return fm, residual.view(batch, 1, n_samples).to(out_dev)
    return fm


def sparse_feature_map_coo(signal, d, n_steps=100):
    """The nonzero entries of sparse_feature_map (:68-125) without the dense [B, A, N] map (4 GiB at the
    headline shape): (flat index b * A * N + a * N + p, value) with repeated picks of one (atom, lag)
    accumulated, sorted by index, plus the map's shape.  No gradient."""
    signal = signal.view(signal.shape[0], 1, -1)
    batch, _, n_samples = signal.shape
    n_atoms = d.shape[0]
    dev = _compute_device(signal)
    with torch.no_grad():
        d_unit = _native.unit_norm(d.detach().to(dev))
        atom, lag, gain, _ = _native.encode_checked(signal.detach().to(dev, torch.float32)[:, 0, :], d_unit, n_steps,
                                                    want_residual=False)
        bidx = torch.arange(batch, device=dev)[:, None].expand_as(atom)
        flat = ((bidx * n_atoms + atom) * n_samples + lag).reshape(-1)
        idx, inv = torch.unique(flat, return_inverse=True)
        val = torch.zeros(idx.shape[0], device=dev).index_add_(0, inv, gain.reshape(-1))
    return idx, val, (batch, n_atoms, n_samples)


def sparse_coding_loss(recon, target, d, n_steps=100, device=None, approx=None, pooling=None):
    """modules/matchingpursuit.py:128-146.  When no gradient is wanted the loss is evaluated on the maps' nonzero
    entries only (every other cell contributes bce(0, 0) = 0), without ever building the dense maps."""
    if not (torch.is_grad_enabled() and recon.requires_grad):
        ri, rv, shape = sparse_feature_map_coo(recon, d, n_steps)
        ti, tv, _ = sparse_feature_map_coo(target, d, n_steps)
        mx = max(rv.max().item(), tv.max().item())
        idx = torch.unique(torch.cat([ri, ti]))
        r = torch.zeros(idx.shape[0], device=rv.device)
        t = torch.zeros(idx.shape[0], device=rv.device)
        r[torch.searchsorted(idx, ri)] = rv / mx
        t[torch.searchsorted(idx, ti)] = tv / mx
        total = F.binary_cross_entropy(r, t, reduction="sum")
        return (total / float(shape[0] * shape[1] * shape[2])).to(recon.device)
    r_map = sparse_feature_map(recon, d, n_steps, device=device, pooling=pooling)
    with torch.no_grad():
        t_map = sparse_feature_map(target, d, n_steps, device=device, pooling=pooling)
    mx = max(r_map.max().item(), t_map.max().item())
    r_map = r_map / mx
    t_map = t_map / mx
    return F.binary_cross_entropy(r_map, t_map)


class _FeatureMapFn(torch.autograd.Function):
    """fm = mp_feature_map_f32(residual, d_unit) (matchingpursuit.py:275-277), differentiable in both arguments.
    The adjoints of the correlation -- a linear convolution of the incoming gradient with the atoms, and the
    correlation of the residual with it -- are tensor operations (fp64 FFTs: exact to fp32 rounding)."""

    @staticmethod
    def forward(ctx, residual, d_unit):
        ctx.save_for_backward(residual, d_unit)
        return _native.feature_map(residual, d_unit)

    @staticmethod
    def backward(ctx, g):
        residual, d_unit = ctx.saved_tensors
        n_samples, atom_size = residual.shape[1], d_unit.shape[1]
        m = n_samples + atom_size
        gs = torch.fft.rfft(g.double(), n=m)  # [B, A, F]
        g_res = g_d = None
        if ctx.needs_input_grad[0]:  # sum_a sum_k g[b, a, n - k] d[a, k]
            ds = torch.fft.rfft(d_unit.double(), n=m)
            g_res = torch.fft.irfft((gs * ds[None]).sum(1), n=m)[:, :n_samples].float()
        if ctx.needs_input_grad[1]:  # sum_b sum_t g[b, a, t] r[b, t + k]
            rs = torch.fft.rfft(residual.double(), n=m)
            g_d = torch.fft.irfft((gs.conj() * rs[:, None]).sum(0), n=m)[:, :atom_size].float()
        return g_res, g_d


def _key_points_with_grad(signal, d, n_steps, dev):
    """The autograd form of sparse_code_to_differentiable_key_points: the reference's graph (:169-224) rebuilt
    around the native encoder's picks -- dense map per step through _FeatureMapFn, value = the map at the pick,
    time through the straight-through softmax of soft_dirac over the per-lag maximum (:192), the residual window
    and the residual update as differentiable gathers / scatters.  Gradients reach the signal and the RAW
    dictionary (through unit_norm, :161), as e_2023_6_7's Encoder trains its dictionary."""
    from .sparse import soft_dirac
    batch, _, n_samples = signal.shape
    n_atoms, atom_size = d.shape
    half = atom_size // 2
    x = signal.to(dev, torch.float32)[:, 0, :]
    dd = d.to(dev, torch.float32)
    d_unit = dd / (torch.norm(dd, dim=-1, keepdim=True) + 1e-8)  # normalization.py:4-6, differentiable
    with torch.no_grad():
        atom, lag, _, _ = _native.encode_checked(x.detach(), _native.unit_norm(dd.detach()), n_steps,
                                                 want_residual=False)
    j = torch.arange(atom_size, device=dev)
    rows = torch.arange(batch, device=dev)
    lin = torch.linspace(0, 1, n_samples, device=dev)
    zero = torch.zeros((), device=dev)
    r = x
    vecs = []
    for i in range(n_steps):
        fm = _FeatureMapFn.apply(r, d_unit)  # [B, A, N]
        a_i, p_i = atom[:, i], lag[:, i]
        value = fm[rows, a_i, p_i]  # :176 (the maximum, at the native pick)
        time = soft_dirac(fm.max(dim=1)[0]) @ lin  # :192
        start = p_i - half
        idx = start[:, None] + j[None, : 2 * half]
        valid = (start[:, None] >= 0) & (idx < n_samples)
        win = torch.where(valid, r.gather(1, idx.clamp(0, n_samples - 1)), zero)  # :199-203
        pad = torch.zeros(batch, atom_size - 2 * half, device=dev)
        vecs.append(torch.cat([value[:, None], time[:, None] * 100, win, pad], dim=1))  # :210-216
        pos = p_i[:, None] + j[None, :]
        upd = torch.where(pos < n_samples, d_unit[a_i] * value[:, None], zero)  # :181, cropped at N
        r = r - torch.zeros_like(r).scatter_add(1, pos.clamp(max=n_samples - 1), upd)  # :223-224
    vecs = torch.stack(vecs, 0).reshape(n_steps * batch, 2 + atom_size)
    return vecs, torch.norm(r, dim=-1).view(batch, 1)


def sparse_code_to_differentiable_key_points(signal, d, n_steps=100, device=None):
    """modules/matchingpursuit.py:149-227: per event the vector
    [value, 100 * time, residual window of `atom_size` samples centred on the event] (the reference
    squeezes that window into `n_atoms` slots, :215, so it needs n_atoms == atom_size), events in step-major,
    batch-minor order, plus the norm of the final residual.  `time` is the argmax of max_a fm over lags on
    linspace(0, 1, N) -- the forward value of the reference's soft_dirac(...) @ linspace (:192).
    The picks come from the native encoder.  Without autograd the windows are cut from the residual of each
    step, replayed from the events with window-sized tensor operations; when the signal or the dictionary
    requires a gradient the reference's graph is rebuilt around the picks (_key_points_with_grad)."""
    signal = signal.view(signal.shape[0], 1, -1)
    batch, _, n_samples = signal.shape
    n_atoms, atom_size = d.shape
    if n_atoms != atom_size:
        raise RuntimeError(f"shape '[{n_atoms}]' is invalid for input of size {atom_size}")  # as :215 raises
    half = atom_size // 2
    out_dev = signal.device
    dev = _compute_device(signal)
    if torch.is_grad_enabled() and (signal.requires_grad or d.requires_grad):
        vecs, rnorm = _key_points_with_grad(signal, d, n_steps, dev)
        return vecs.to(out_dev), rnorm.to(out_dev)
    with torch.no_grad():
        x = signal.detach().to(dev, torch.float32)[:, 0, :]
        d_unit = _native.unit_norm(d.detach().to(dev))
        atom, lag, gain, residual = _native.encode_checked(x, d_unit, n_steps)
        r = x.clone()
        j = torch.arange(atom_size, device=dev)
        rows = torch.arange(batch, device=dev)
        lin = torch.linspace(0, 1, n_samples, device=dev)
        vecs = torch.zeros(n_steps, batch, 2 + atom_size, device=dev)
        for i in range(n_steps):
            p = lag[:, i]
            # residual[j, 0, pos - half: pos + half] with python slice semantics (:199): a negative start
            # wraps to the end of the array, which leaves an empty (zero-filled) window
            start = p - half
            idx = start[:, None] + j[None, : 2 * half]
            valid = (start[:, None] >= 0) & (idx < n_samples)
            win = torch.where(valid, r.gather(1, idx.clamp(0, n_samples - 1)), torch.zeros((), device=dev))
            vecs[i, :, 0] = gain[:, i]
            vecs[i, :, 1] = lin[p] * 100
            vecs[i, :, 2: 2 + 2 * half] = win
            # r -= gain * atom at lag, cropped at N
            pos = p[:, None] + j[None, :]
            ok = pos < n_samples
            upd = torch.where(ok, d_unit[atom[:, i]] * gain[:, i, None], torch.zeros((), device=dev))
            r.scatter_add_(1, pos.clamp(max=n_samples - 1), -upd)
        return (vecs.reshape(n_steps * batch, 2 + atom_size).to(out_dev),
                torch.norm(residual, dim=-1).view(batch, 1).to(out_dev))


class SparseCodingLoss(nn.Module):
    """modules/matchingpursuit.py:422-463."""

    def __init__(self, n_atoms, atom_size, n_steps, approx, learning_steps, device=None, pooling=None):
        super().__init__()
        self.approx = approx
        self.n_steps = n_steps
        self.learning_steps = learning_steps
        self._steps_executed = 0
        self.d = unit_norm(torch.zeros(n_atoms, atom_size, device=device).uniform_(-1, 1))
        self.pooling = pooling

    def _learning_step(self, signal):
        with torch.no_grad():
            self.d[:] = dictionary_learning_step(
                signal, self.d, n_steps=self.n_steps, device=signal.device, approx=self.approx).to(self.d.device)
            self._steps_executed += 1

    def loss(self, recon, target):
        if self._steps_executed < self.learning_steps:
            self._learning_step(target)
        return sparse_coding_loss(recon, target, self.d, n_steps=self.n_steps, device=recon.device,
                                  pooling=self.pooling)

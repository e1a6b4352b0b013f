#!/usr/bin/env python3
"""Would a LAZY screen pay?  Simulation (exact feature maps from mp_feature_map_f32, no kernel changes).

Idea: after an event (atom a*, gain g, lag p) the cells of the dirty lag range change by at most g * mu(a*, tile),
mu = max |<d_a*, shifted d_a>| over the tile's atoms and all shifts (coherence).  A tile whose dirty cells all satisfy
    approx + eps + g * mu(a*, tile)  <  LB0        (LB0 = best lower bound over the cells the event cannot touch)
cannot hold the next maximum: skip its 16 transforms, keep its keys, add g * mu to their eps (still valid bounds).
The exact refinement and the select stay as they are, so the result stays bit-exact.  This script counts, per step,
how many (segment, tile) pairs would be skipped and how many cells would contend, with the bounds carried along
exactly as a kernel would carry them.   lazy_screen_sim.py [B] [K]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth
A, L, N = 512, 512, 32768
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
POLICY = int(sys.argv[3]) if len(sys.argv) > 3 else 0
TAU = 2e-5
d = synth.make_dictionary(A, L, seed=1000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
NAT, NBLK = A // 32, N // 64
# coherence of every atom with every tile: max over the tile's atoms and all shifts of |cross-correlation|
Df = torch.fft.rfft(torch.nn.functional.pad(du, (0, L)), dim=-1)                       # [A, L+1]
mu = torch.zeros(A, NAT, device="cuda")
for a0 in range(0, A, 64):
    cc = torch.fft.irfft(Df[a0:a0 + 64, None, :] * torch.conj(Df)[None], n=2 * L, dim=-1).abs().amax(dim=-1)  # [64, A]
    mu[a0:a0 + 64] = cc.view(-1, NAT, 32).amax(dim=-1)
mu = mu * 1.0001 + 1e-6
print(f"coherence atom-vs-tile: mean {float(mu.mean()):.3f} max off-own-tile {float(mu.masked_fill(torch.eye(A, device='cuda').view(A, NAT, 32).amax(-1) > 0, 0).max()):.3f}")
atom, lag, gain, _ = nat.encode(x, du, K, path=nat.MP_PATH_INCREMENTAL)
res = x.clone()
def cells_of(r):
    fm = nat.feature_map(r, du)                                     # [B, A, N] exact
    return fm.view(B, NAT, 32, NBLK, 64).amax(dim=(2, 4)).permute(0, 2, 1).contiguous()   # [B, NBLK, NAT]
approx = cells_of(res)                                              # full pass: all fresh
wn = torch.sqrt(torch.nn.functional.avg_pool1d((res ** 2)[:, None], 2048, 64, padding=0, ceil_mode=True)[:, 0] * 2048)
eps = torch.full_like(approx, 0.0) + TAU * float(res.norm(dim=-1).max())   # a generous per-cell eps
j = torch.arange(L, device="cuda")
skipped_hist, cont_hist, stale_hist = [], [], []
for k in range(K):
    # contenders with the bounds as carried
    lb = (approx - eps).amax(dim=(1, 2))
    cont = ((approx + eps) >= lb[:, None, None]).sum(dim=(1, 2))
    cont_hist.append(cont.float().mean().item())
    a, p, g = atom[:, k], lag[:, k], gain[:, k]
    pos = p[:, None] + j[None]
    ok = pos < N
    res.scatter_add_(1, pos.clamp(max=N - 1), torch.where(ok, -g[:, None] * du[a], torch.zeros((), device="cuda")))
    fresh = cells_of(res)
    fb = ((p - L + 1).clamp(min=0) // 64); lbk = ((p + L - 1).clamp(max=N - 1) // 64)
    blk = torch.arange(NBLK, device="cuda")[None, :]
    dirty = (blk >= fb[:, None]) & (blk <= lbk[:, None])                    # [B, NBLK]
    lb0 = (approx - eps).masked_fill(dirty[:, :, None], -1e30).amax(dim=(1, 2))          # cells the event cannot touch
    slack = g[:, None].abs() * mu[a]                                        # [B, NAT]
    ub_dirty = (approx + eps).masked_fill(~dirty[:, :, None], -1e30).amax(dim=1)          # [B, NAT]
    skip = (ub_dirty + slack) < lb0[:, None]                                # [B, NAT]
    if POLICY >= 1:  # a tile with an already-stale dirty cell is recomputed (staleness never stacks)
        has_stale = ((eps > 1e-3) & dirty[:, :, None]).any(dim=1)
        skip = skip & ~has_stale
    if POLICY >= 2:  # ... and only skip with a safety margin against the decay of the maximum
        skip = skip & ((ub_dirty + slack) < 0.7 * lb0[:, None])
    skipped_hist.append(skip.float().mean().item())
    upd = dirty[:, :, None] & ~skip[:, None, :]
    stl = dirty[:, :, None] & skip[:, None, :]
    approx = torch.where(upd, fresh, approx)
    eps = torch.where(upd, torch.full_like(eps, TAU * float(res.norm(dim=-1).max())), eps)
    eps = torch.where(stl, eps + slack[:, None, :], eps)
    stale_hist.append(((eps > 1e-3).float().mean().item()))
    # sanity: bounds stay valid
    assert bool(((fresh <= approx + eps + 1e-6) & (fresh >= approx - eps - 1e-6)).all()), "bound violated"
    if k % 8 == 7 or k == K - 1:
        print(f"step {k + 1:3d}: tiles skipped {np.mean(skipped_hist[-8:]) * 100:5.1f} %   contenders per segment {np.mean(cont_hist[-8:]):6.2f}   cells with inflated eps {stale_hist[-1] * 100:5.2f} %", flush=True)
print(f"overall: {np.mean(skipped_hist) * 100:.1f} % of (segment, tile) screens skippable; contenders mean {np.mean(cont_hist):.2f} max-step-mean {np.max(cont_hist):.2f}")

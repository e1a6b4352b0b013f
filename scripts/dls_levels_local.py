#!/usr/bin/env python3
"""The multi-rank form of dictionary_learning_step (by dependency levels, two launches per level around the all-reduce:
mpcore/matchingpursuit.py::_dictionary_update_by_levels) run in ONE process, where the all-reduce is the identity: what the
form costs on the device and in Python without any transport -- beside the single-process level kernel and beside the
two-rank gloo rehearsal of scripts/dls_dist_time.py (whose all-reduces are CPU round trips).  python scripts/dls_levels_local.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import matchingpursuit as mpm, _native as nat, synth
A, L, N, B, K = 512, 512, 32768, 64, 64
dn = synth.make_dictionary(A, L, seed=1000)
d = torch.from_numpy(dn).cuda()
x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=3 * K, seed=1002)).cuda()


def by_levels(sig, collect=None):
    d_work = nat.unit_norm(d)
    residual = sig.clone()
    atom, lag, gain, _ = nat.encode_checked(sig, d_work, K, want_residual=False)
    if collect is not None:
        torch.cuda.synchronize(); collect.append(time.perf_counter())
    rows = d_work[atom] * gain[..., None]
    anorm = torch.norm(rows, dim=-1)
    order = mpm.first_selection_order(atom.cpu().numpy())
    return mpm._dictionary_update_by_levels(residual, d_work, atom, lag, rows, anorm, order, None)


for name, fn in (("single-process level kernel", lambda: mpm.dictionary_learning_step(x[:, None, :], d, n_steps=K)),
                 ("multi-rank form, one process (all-reduce = identity), 64 segments", lambda: by_levels(x)),
                 ("multi-rank form, one process, 32 segments (one rank's share of two)", lambda: by_levels(x[:32]))):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per step", flush=True)
marks = []
t0 = time.perf_counter(); by_levels(x[:32], marks); torch.cuda.synchronize()
print(f"  of which unit_norm + encode of 32 segments: {(marks[0] - t0) * 1e3:.2f} ms", flush=True)
ref = mpm.dictionary_learning_step(x[:, None, :], d, n_steps=K)
print("max |by levels - level kernel| =", float((by_levels(x) - ref).abs().max()), flush=True)

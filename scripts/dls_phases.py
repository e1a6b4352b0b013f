#!/usr/bin/env python3
"""Where dictionary_learning_step's time goes beside its encode: the single-device body of
mpcore/matchingpursuit.py::dictionary_learning_step replayed phase by phase with a device synchronisation after each
(so the phases add up to MORE than the real call, which overlaps host and device work).  python scripts/dls_phases.py [K] [B]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import matchingpursuit as mp, _native as nat, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
A, L, N = 512, 512, 32768
dn = synth.make_dictionary(A, L, seed=1000)
d = torch.from_numpy(dn).cuda()
x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=3 * K, seed=1002)).cuda()
sig3 = x[:, None, :]
for _ in range(4):
    mp.dictionary_learning_step(sig3, d, n_steps=K)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    mp.dictionary_learning_step(sig3, d, n_steps=K)
torch.cuda.synchronize()
print(f"K {K} B {B}: dictionary_learning_step {(time.perf_counter() - t0) * 100:.2f} ms per call", flush=True)
acc = {}
def mark(name, t):
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t) * 1e3
    return time.perf_counter()
R = 10
for _ in range(R):
    t = time.perf_counter()
    d_work = nat.unit_norm(d); residual = x.clone(); t = mark("unit_norm + clone", t)
    atom, lag, gain, _ = nat.encode_checked(x, d_work, K, want_residual=False); t = mark("encode_checked", t)
    rows = d_work[atom] * gain[..., None]; anorm = torch.norm(rows, dim=-1); t = mark("rows, norms", t)
    atom_h = atom.cpu().numpy()
    order = mp.first_selection_order(atom_h); t = mark("first_selection_order", t)
    perm, counts = mp.group_events_by_atom(atom_h, order, A); t = mark("group_events_by_atom", t)
    perm_h = perm.numpy()
    perm_d = perm.cuda(); ev_batch = perm_d // K; ev_lag = lag.reshape(-1)[perm_d]
    ev_rows = rows.reshape(-1, L)[perm_d]; ev_norm = anorm.reshape(-1)[perm_d]; t = mark("device gathers", t)
    offsets_h = np.zeros(len(order) + 1, dtype=np.int64); offsets_h[1:] = np.cumsum(np.asarray(counts, dtype=np.int64))
    host = (offsets_h, perm_h // K, lag.cpu().numpy().reshape(-1)[perm_h]); t = mark("host arrays", t)
    nl = nat.dictionary_update(residual, d_work, torch.from_numpy(np.asarray(order, dtype=np.int64)), torch.from_numpy(offsets_h), ev_batch, ev_lag,
                               ev_rows.contiguous(), ev_norm.contiguous(), host_events=host); t = mark("dictionary_update (levels + launches)", t)
    out = nat.unit_norm(d_work); t = mark("final unit_norm", t)
print({k: round(v / R, 3) for k, v in acc.items()}, "levels", nl, "atoms used", len(order), "sum", round(sum(acc.values()) / R, 2), flush=True)

#!/usr/bin/env python3
"""Sweep of the persistent form's sizing knobs at the headline shape, settings alternated so that box and clock drift
cancel: select workers (MP_TUNE_PERSIST_SELECTS), ticket shards (MP_TUNE_PERSIST_SHARDS), workgroups
(MP_TUNE_PERSIST_WORKERS).  python scripts/persist_sweep.py [B]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
A, L, N, K = 512, 512, 32768, 64
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT); torch.cuda.synchronize()
for _ in range(30):
    nat.encode(x, du, K, path=nat.MP_PATH_FFT)
torch.cuda.synchronize()
def sweep(key, name, values):
    times = {v: [] for v in values}
    for rep in range(5):
        for v in values:
            nat.tune(key, v)
            out = nat.encode(x, du, K, path=nat.MP_PATH_FFT); torch.cuda.synchronize()
            assert all(torch.equal(p, q) for p, q in zip(out, ref)), (name, v)
            t0 = time.perf_counter()
            for _ in range(10):
                nat.encode(x, du, K, path=nat.MP_PATH_FFT)
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) / 10)
    nat.tune(key, 0)
    print(name, "  ".join(f"{v}: {np.median(times[v]) * 1e3:.3f} ms" for v in values), flush=True)
which = sys.argv[2] if len(sys.argv) > 2 else "all"
if which in ("all", "selects"):
    sweep(8, "select workers (0 = table)", [0, 24, 32, 40, 48, 56, 64, 80, 96])
if which in ("all", "shards"):
    sweep(6, "ticket shards (0 = heuristic)", [0, 1, 2, 4, 8])
if which in ("all", "workers"):
    sweep(7, "workgroups (0 = 768)", [0, 640, 704, 736])

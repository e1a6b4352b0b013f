#!/usr/bin/env python3
"""Sweep of the lazy screen's margin (MP_TUNE_LAZY_MARGIN) inside the persistent launch at the headline shape, on the three
signals bench.py times (3 K, K / 2 and no planted events per segment), settings alternated so that drift cancels; results
must be identical at every value, segments marked as overflow are counted.  python scripts/lazy_knob_sweep.py [B [K]]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
A, L, N = int(os.environ.get('SWEEP_A', 512)), int(os.environ.get('SWEEP_L', 512)), int(os.environ.get('SWEEP_N', 32768))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
values = [0.7, 0.85, 0.9, 0.95]
for name, n_ev, seed in (("3K planted", 3 * K, 1002), ("K/2 planted", K // 2, 2002), ("unplanted", 0, 2002)):
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=n_ev, seed=seed)).cuda()
    nat.tune(10, 0)
    ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT); torch.cuda.synchronize()
    for _ in range(20): nat.encode(x, du, K, path=nat.MP_PATH_FFT)
    torch.cuda.synchronize()
    times = {v: [] for v in values}; marks = {}
    for rep in range(4):
        for v in values:
            nat.tune(10, v)
            out = nat.encode(x, du, K, path=nat.MP_PATH_FFT); torch.cuda.synchronize()
            nanrows = torch.isnan(out[2]).any(dim=1)
            marks[v] = int(nanrows.sum())
            ok = ~nanrows & ~torch.isnan(ref[2]).any(dim=1)
            assert torch.equal(out[0][ok], ref[0][ok]) and torch.equal(out[1][ok], ref[1][ok]) and torch.equal(out[2][ok], ref[2][ok]), (name, v)
            t0 = time.perf_counter()
            for _ in range(10): nat.encode(x, du, K, path=nat.MP_PATH_FFT)
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) / 10)
    nat.tune(10, 0)
    print(f"B{B} K{K} {name:12s}", "  ".join(f"{v}: {np.median(times[v]) * 1e3:.3f} ms ({marks[v]} marked)" for v in values), flush=True)

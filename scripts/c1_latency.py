#!/usr/bin/env python3
"""BASELINE configs[0] (16 x 256 dictionary, one 8192-sample segment, 8 iterations): latency of one encode per
schedule -- the small end, where launches rather than arithmetic set the time."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth
A, L, N, B, K = 16, 256, 8192, 1, 8
d = synth.make_dictionary(A, L, seed=100)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=24, seed=101)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
for name, path in (("fft", nat.MP_PATH_FFT), ("incremental", nat.MP_PATH_INCREMENTAL), ("direct", nat.MP_PATH_DIRECT)):
    ts = []
    for r in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = nat.encode(x, du, K, path=path, want_residual=False); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    print(f"c1 {name:12s}: median {np.median(ts[5:]):7.1f} us per encode ({B * K / np.median(ts[5:]) * 1e6:8.0f} segment-iterations/s)", flush=True)

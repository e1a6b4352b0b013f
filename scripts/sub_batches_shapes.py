#!/usr/bin/env python3
"""Is the sub-batch default (four, from 48 segments) right away from the headline shape?  The multiband model's
band shapes at 64 segments: one stream against the default."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth
B, K, A = 64, 32, 1024
for N in (512, 1024, 2048, 4096, 8192, 16384):
    L = N // 4
    d = synth.make_dictionary(A, L, seed=N)
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=48, seed=N)).cuda()
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    row = []
    for name, flags in (("one stream", nat.MP_FLAG_NO_OVERLAP), ("default", 0)):
        f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
        f(); f(); torch.cuda.synchronize(); ts = []
        for _ in range(7):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        row.append(f"{name} {float(np.median(ts)) * 1e3:7.2f} ms")
    print(f"N{N:6d} L{L:5d} A{A} B{B} K{K}: " + " | ".join(row), flush=True)

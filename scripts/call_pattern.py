#!/usr/bin/env python3
"""One stream or two sub-batches on forked streams?  It depends on how the caller calls: encodes queued back to
back (a throughput loop: bench.py) against one encode at a time with a synchronisation after each (what a drop-in
user of sparse_code does).  Headline dictionary, several batch sizes."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth
A, L, N, K = 512, 512, 32768, 64
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
for B in (64, 128, 256):
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
    row = []
    for name, flags in (("one stream", nat.MP_FLAG_NO_OVERLAP), ("two sub-batches", nat.MP_FLAG_OVERLAP)):
        f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
        f(); f(); torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        one = float(np.median(ts))
        t0 = time.perf_counter()
        for _ in range(6): f()
        torch.cuda.synchronize(); back = (time.perf_counter() - t0) / 6
        row.append(f"{name}: one at a time {B * K / one:8.0f}, back to back {B * K / back:8.0f}")
    print(f"B{B:4d}: " + " | ".join(row) + "  segment-iterations/s", flush=True)

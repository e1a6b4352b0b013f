#!/usr/bin/env python3
"""What do the profiler's hipEvents cost?  Headline shape, MP_PATH_FFT / INCREMENTAL, events on vs off."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth
A, L, N, B, K = 512, 512, 32768, 64, 64
d = synth.make_dictionary(A, L, seed=1000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
for path in (nat.MP_PATH_FFT, nat.MP_PATH_INCREMENTAL):
    for on in (False, True, False, True):
        nat.profile_enable(on)
        ts = []
        for r in range(6):
            torch.cuda.synchronize(); nat.profile_read()
            t0 = time.perf_counter()
            nat.encode(x, du, K, path=path, want_residual=False)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"path {path} events {'on ' if on else 'off'}: median {np.median(ts[1:]):.3f} ms  min {min(ts[1:]):.3f}", flush=True)
nat.profile_enable(False)

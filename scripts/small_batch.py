#!/usr/bin/env python3
"""Where the FFT schedule starts to pay: headline dictionary (512 x 512, N = 32768, K = 64), batch 1..64,
FFT vs incremental (the FFT schedule wins from one segment up with this dictionary)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth
A, L, N, K = 512, 512, 32768, 64
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
xall = torch.from_numpy(synth.make_segments(64, N, d, n_events=192, seed=1002)).cuda()
for B in (1, 2, 4, 8, 16, 32, 64):
    x = xall[:B].contiguous()
    row = []
    for path in (nat.MP_PATH_FFT, nat.MP_PATH_INCREMENTAL):
        ts = []
        for r in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            nat.encode(x, du, K, path=path, want_residual=False); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        row.append(np.median(ts[1:]))

    print(f"B {B:3d}: fft {row[0]:7.3f} ms  incremental {row[1]:7.3f} ms", flush=True)

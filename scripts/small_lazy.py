#!/usr/bin/env python3
"""Is the lazy screen worth its table below 24 segments (where the Python mirror did not ask for it)?  Persistent form with
and without the table, 1 .. 24 segments, two dictionaries."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
for A, L, N, K in ((512, 512, 32768, 64), (1024, 1024, 32768, 32), (64, 300, 8192, 16), (16, 256, 8192, 8)):
    d = synth.make_dictionary(A, L, seed=A + L)
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    mu = nat.coherence_table(du)
    for B in (1, 2, 4, 8, 16, 24):
        x = torch.from_numpy(synth.make_segments(B, N, d, n_events=3 * K, seed=7)).cuda()
        row = f"{A}x{L} K{K} B{B:3d}:"
        for name, co in (("plain", False), ("lazy", mu)):
            f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=co)
            out = f(); out = f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8): out = f()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
            row += f"  {name} {dt * 1e3:6.3f} ms ({B * K / dt / 1e3:5.0f} k, schedule {nat.last_schedule()})"
        print(row, flush=True)

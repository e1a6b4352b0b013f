#!/usr/bin/env python3
"""Shapes between the headline's 8192 cells per segment and the config-4 shape's 262144: larger dictionaries and longer
segments with 1024- to 4096-point transforms.  Which form the default takes there, and what it does per second."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
shapes = [(2048, 512, 32768, 64, 64), (1024, 512, 65536, 32, 64), (512, 512, 131072, 32, 64), (1024, 1024, 32768, 64, 32),
          (4096, 256, 16384, 64, 32)]
for A, L, N, B, K in shapes:
    d = synth.make_dictionary(A, L, seed=A + L)
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=3 * K, seed=7)).cuda()
    ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_NO_OVERLAP, coherence=False)
    torch.cuda.synchronize()
    cells = ((N + 63) // 64) * ((A + 31) // 32)
    line = f"{A}x{L}, {B} x {N}, K={K} ({cells} cells per segment):"
    for name, kw in (("one stream", dict(flags=nat.MP_FLAG_NO_OVERLAP, coherence=False)), ("default, no table", dict(coherence=False)),
                     ("default", dict())):
        f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, **kw)
        out = f(); out = f(); out = f(); torch.cuda.synchronize()
        same = all(torch.equal(p, q) for p, q in zip(out, ref))
        t0 = time.perf_counter()
        for _ in range(4): out = f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
        line += f"  {name}: {dt * 1e3:.2f} ms = {B * K / dt / 1e3:.0f} k (schedule {nat.last_schedule()}, identical {same}, marked {int(torch.isnan(out[2]).any(dim=1).sum())})"
    print(line, flush=True)

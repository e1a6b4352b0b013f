#!/usr/bin/env python3
"""Per-call time of the library default (coherence=None: the content-validated table cache decides) on a mid-size shape."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
A, L, N, B, K = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (2048, 512, 32768, 64, 64)
d = synth.make_dictionary(A, L, seed=A + L)
du = nat.unit_norm(torch.from_numpy(d).cuda())
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=3 * K, seed=7)).cuda()
for co_name, kw in (("coherence=False", dict(coherence=False)), ("coherence=None (auto)", dict())):
    nat.clear_caches()
    ts = []
    for call in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, **kw)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ls = nat.lazy_stats()
    print(co_name, "per call ms:", [round(t, 2) for t in ts], "schedule", nat.last_schedule(), ls, flush=True)

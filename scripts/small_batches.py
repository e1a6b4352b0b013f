#!/usr/bin/env python3
"""Few segments: the library default against the persistent form (with and without the lazy screen) and the one-stream
launch-per-step form.  python scripts/small_batches.py [A L N]"""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
A, L, N, K = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 64) if len(sys.argv) > 3 else (512, 512, 32768, 64)
d = synth.make_dictionary(A, L, seed=1000); du = nat.unit_norm(torch.from_numpy(d).cuda()); mu = nat.coherence_table(du)
def rate(B, x, **kw):
    f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, **kw)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): f()
    torch.cuda.synchronize(); return B * K * 8 / (time.perf_counter() - t0) / 1e3
for B in (1, 2, 3, 4, 6, 8, 12):
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
    row = [f"default {rate(B, x, coherence=False):6.0f} k (schedule {nat.last_schedule()})"]
    row.append(f"persistent {rate(B, x, flags=nat.MP_FLAG_FFT_PERSISTENT, coherence=False):6.0f} k")
    row.append(f"persistent + lazy {rate(B, x, flags=nat.MP_FLAG_FFT_PERSISTENT, coherence=mu):6.0f} k")
    row.append(f"one stream {rate(B, x, flags=nat.MP_FLAG_NO_OVERLAP, coherence=False):6.0f} k")
    print(f"B{B:3d}: " + " | ".join(row), flush=True)

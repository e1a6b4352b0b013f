#!/usr/bin/env python3
"""BASELINE configs[3] at full size on the library default, one mp_tune knob swept at a time: sub-batches (MP_TUNE_GROUPS),
pairs per slot of the screen (MP_TUNE_SCREEN_PPS), widenings a bound may take (MP_TUNE_LAZY_REUSE)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
A, L, N, K, B = 4096, 2048, 131072, 256, 128
d = synth.make_dictionary(A, L, seed=4000)
x = torch.empty(B, N, device="cuda")
for b0 in range(0, B, 32):
    x[b0:b0 + 32] = torch.from_numpy(synth.make_segments(32, N, d, n_events=3 * K, seed=4001, first_index=b0)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
mu = nat.coherence_table(du)
ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=mu); torch.cuda.synchronize()
def sweep(key, name, values):
    row = name + ":"
    for v in values:
        nat.tune(key, v)
        out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=mu); torch.cuda.synchronize()
        ok = all(torch.equal(p, q) for p, q in zip(out, ref))
        t0 = time.perf_counter()
        out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=mu); out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=mu)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
        row += f"  {v}: {dt*1e3:.1f} ms{'' if ok else ' MISMATCH'}"
    nat.tune(key, reset)
    print(row, flush=True)
reset = 4
sweep(3, "sub-batches (default 4)", [4, 2, 3, 4])
reset = 0
sweep(2, "screen pps (0 = heuristic)", [0, 2, 4, 8, 16])
sweep(12, "lazy reuse (0 = 4)", [0, 1, 2, 3])

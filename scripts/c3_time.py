#!/usr/bin/env python3
"""BASELINE configs[3] at full size on the library default (sub-batches on forked streams) and on one stream: wall time of two
encodes after two warm-ups.  `python scripts/c3_time.py once`: one default encode after one warm-up (for a kernel trace: how
far do the sub-batches' selects overlap the other sub-batches' screens?)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "matching-pursuit_amd")
from mpcore import _native as nat, synth
A, L, N, K, B = 4096, 2048, 131072, 256, 128
d = synth.make_dictionary(A, L, seed=4000)
x = torch.empty(B, N, device="cuda")
for b0 in range(0, B, 32):
    x[b0:b0 + 32] = torch.from_numpy(synth.make_segments(32, N, d, n_events=3 * K, seed=4001, first_index=b0)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
if len(sys.argv) > 1 and sys.argv[1] == "once":
    nat.encode(x, du, K, path=nat.MP_PATH_FFT); torch.cuda.synchronize()
    nat.encode(x, du, K, path=nat.MP_PATH_FFT); torch.cuda.synchronize()
    sys.exit(0)
for name, flags in (("default (4 sub-batches)", 0), ("one stream", nat.MP_FLAG_NO_OVERLAP)):
    out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags); out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags); out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print(f"configs[3] {name}: {dt*1e3:.1f} ms = {B*K/dt/1e3:.1f} k (schedule {nat.last_schedule()})", flush=True)

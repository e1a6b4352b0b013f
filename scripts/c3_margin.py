#!/usr/bin/env python3
"""BASELINE configs[3] at full size on the library default, lazy margin of the launch-per-step form swept
(MP_TUNE_LAZY_MARGIN; 0 = the table's 0.85)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "matching-pursuit_amd")
from mpcore import _native as nat, synth
A, L, N, K, B = [int(os.environ.get(k, v)) for k, v in (("SWEEP_A", 4096), ("SWEEP_L", 2048), ("SWEEP_N", 131072), ("SWEEP_K", 256), ("SWEEP_B", 128))]
NEV = int(os.environ.get("SWEEP_EVENTS", 3 * K))
d = synth.make_dictionary(A, L, seed=4000)
x = torch.empty(B, N, device="cuda")
for b0 in range(0, B, 32):
    x[b0:b0 + 32] = torch.from_numpy(synth.make_segments(32, N, d, n_events=NEV, seed=4001, first_index=b0)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
mu = nat.coherence_table(du)
ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=mu); torch.cuda.synchronize()
for v in [float(a) for a in sys.argv[1:]] or [0, 0.9, 0.95, 1.0, 0]:
    nat.tune(10, v)
    out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=mu); torch.cuda.synchronize()
    ok = all(torch.equal(p, q) for p, q in zip(out, ref))
    t0 = time.perf_counter()
    out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=mu); out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=mu)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print(f"{A}x{L} B{B} N{N} K{K} events {NEV} margin {v}: {dt*1e3:.1f} ms = {B*K/dt/1e3:.1f} k  identical {ok}  marked {int(torch.isnan(out[2]).any(dim=1).sum())}", flush=True)
nat.tune(10, 0)

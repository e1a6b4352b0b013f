#!/usr/bin/env python3
"""Where a long-atom encode's time goes (the multiband model's two largest bands: 1024 atoms of 4096 / 8192 samples,
bands of 16384 / 32768 samples, 32 steps): per-kind launch spans (mp_profile_*), segments marked, on planted events and
on plain noise.   python scripts/long_atom_probe.py [batch]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # e.g. MP_FLAG_REFINE_MFMA
print("flags", FLAGS, flush=True)
K = 32
for N in (8192, 16384, 32768):
    A, L = 1024, N // 4
    d = synth.make_dictionary(A, L, seed=N)
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    rng = np.random.default_rng(N)
    for kind, xh in (("planted", synth.make_segments(B, N, d, n_events=48, seed=N)),
                     ("noise", rng.standard_normal((B, N)).astype(np.float32))):
        x = torch.from_numpy(xh).cuda()
        nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=FLAGS); torch.cuda.synchronize()
        for every in (0, 1):
            nat.profile_enable(every); nat.profile_read()
            t0 = time.perf_counter()
            out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=FLAGS)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            p = nat.profile_read()
            if every:
                print(f"N{N} L{L} B{B} {kind}: {dt0 * 1e3:.2f} ms ({dt * 1e3:.2f} with events), schedule {nat.last_schedule()}, marked "
                      f"{int(torch.isnan(out[2]).any(dim=1).sum())}; spans",
                      {k: (round(v[0] / max(v[1], 1), 3), v[1]) for k, v in p.items()}, flush=True)
            dt0 = dt
        nat.profile_enable(0)

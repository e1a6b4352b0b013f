#!/usr/bin/env python3
"""Interleaved A/B of mp_encode_f32 flag sets at the headline shape (one process, N rounds,
median and min per variant -- cdna_hip_programming.md section 5.4 rule 24)."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat  # noqa: E402
from mpcore import synth  # noqa: E402

A, L, N, B, K = 512, 512, 32768, 64, 64
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
variants = [("inc", 2, 0), ("inc_nodma", 2, 1), ("inc_nostagger", 2, 16), ("inc_nostagger_nodma", 2, 17),
            ("inc_ta64", 2, 4), ("inc_ta64_nostagger", 2, 20), ("inc_np", 2, 8),
            ("direct", 0, 0), ("direct_nostagger", 0, 16), ("direct_ta64", 0, 4)]
d = synth.make_dictionary(A, L, seed=1000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
times = {v[0]: [] for v in variants}
for r in range(rounds + 1):
    for name, path, flags in variants:
        k = K if path == 2 else 8
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nat.encode(x, du, k, path=path, flags=flags, want_residual=False)
        torch.cuda.synchronize()
        if r > 0:
            times[name].append((time.perf_counter() - t0) * 1e3)
for name, path, flags in variants:
    t = np.array(times[name])
    k = K if path == 2 else 8
    print(f"{name:20s} median {np.median(t):8.3f} ms  min {t.min():8.3f} ms  -> {B * k / np.median(t) * 1e3:9.0f} seg-it/s", flush=True)

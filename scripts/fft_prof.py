#!/usr/bin/env python3
"""Run MP_PATH_FFT (or another path) a few times at a chosen shape, for rocprofv3 --kernel-trace."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat  # noqa: E402
from mpcore import synth  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
path = int(sys.argv[2]) if len(sys.argv) > 2 else nat.MP_PATH_FFT
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
A, L, N, B, K = {"c2": (512, 512, 32768, 64, 64), "c4": (4096, 2048, 131072, 16, 16)}[shape]
d = synth.make_dictionary(A, L, seed=1000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=min(3 * K, 192), seed=1002)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
nat.encode(x, du, 2, path=path)
torch.cuda.synchronize()
for r in range(reps):
    t0 = time.perf_counter()
    out = nat.encode(x, du, K, path=path, want_residual=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{shape} path {path}: {dt * 1e3:.2f} ms -> {B * K / dt:.0f} seg-it/s nan={torch.isnan(out[2]).any().item()}", flush=True)

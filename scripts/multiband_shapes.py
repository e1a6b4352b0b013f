"""Encode rate at the shapes of the reference's multiband model (experiments/archive/e_2023_3_8/experiment.py:
351-359: 1024 atoms per band, band sizes 512 ... 32768 samples, atoms a quarter of the band, 32 steps).
Usage: python scripts/multiband_shapes.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = 32
for N in (512, 1024, 2048, 4096, 8192, 16384, 32768):
    A, L = 1024, N // 4
    d = synth.make_dictionary(A, L, seed=N)
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=48, seed=N)).to(DEV)
    du = nat.unit_norm(torch.from_numpy(d).to(DEV))
    row = []
    for name, path in (("default", nat.default_path(L)), ("incremental", nat.MP_PATH_INCREMENTAL)):
        if name == "incremental" and path == nat.default_path(L):
            continue
        try:
            nat.encode(x, du, K, path=path)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                out = nat.encode(x, du, K, path=path)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            row.append(f"{name}(path {path}) {dt * 1e3:8.2f} ms {B * K / dt:9.0f} seg-it/s")
        except Exception as e:  # noqa: BLE001
            row.append(f"{name}: {e}")
    print(f"N{N:6d} L{L:5d} A{A} B{B} K{K}: " + " | ".join(row), flush=True)

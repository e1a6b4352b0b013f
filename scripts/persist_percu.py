#!/usr/bin/env python3
"""Persistent form: workgroups per CU (2 / 3 / 4) against the batch size -- 8 encodes in a row, headline dictionary.
LAZY=1 in the environment: with the lazy screen (the coherence table passed explicitly)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth
A, L, N, K = 512, 512, 32768, 64
if len(sys.argv) > 1:
    A, L, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
MU = nat.coherence_table(du) if os.environ.get('LAZY') else False   # LAZY=1: the persistent form with the lazy screen
def rate(B, x, flags, n=8):
    f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags, coherence=(MU if flags == nat.MP_FLAG_FFT_PERSISTENT else False))
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return B * K * n / (time.perf_counter() - t0) / 1e3


mode = sys.argv[4] if len(sys.argv) > 4 else "percu"
if mode == "percu":     # workgroups per CU against the batch size, and the launch-per-step forms beside them
    for B in (16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 112, 128):
        x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
        row = []
        for pcu in (2, 3):
            nat.tune(nat.MP_TUNE_PERSIST_WORKERS, 256 * pcu)
            row.append(f"{pcu}/CU {rate(B, x, nat.MP_FLAG_FFT_PERSISTENT):6.0f} k")
        nat.tune(nat.MP_TUNE_PERSIST_WORKERS, 0)
        row.append(f"heuristic {rate(B, x, nat.MP_FLAG_FFT_PERSISTENT):6.0f} k")
        row.append(f"per step, one stream {rate(B, x, nat.MP_FLAG_NO_OVERLAP):6.0f} k, sub-batches {rate(B, x, nat.MP_FLAG_FFT_NO_PERSISTENT):6.0f} k")
        print(f"B{B:4d}: " + " | ".join(row), flush=True)
else:                   # one batch size: total workgroups x select workers
    B = int(mode)
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
    for nsel in (32, 48, 64, 96, 128):
        nat.tune(nat.MP_TUNE_PERSIST_SELECTS, nsel)
        row = []
        for workers in (512, 576, 640, 704, 768):
            nat.tune(nat.MP_TUNE_PERSIST_WORKERS, workers)
            row.append(f"{workers}: {rate(B, x, nat.MP_FLAG_FFT_PERSISTENT):6.0f} k")
        print(f"B{B} select workers {nsel}: " + " | ".join(row), flush=True)

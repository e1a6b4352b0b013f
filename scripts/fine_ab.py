#!/usr/bin/env python3
"""MP_TUNE_PERSIST_FINE 1 / 2 (one / two slots per tile quarter) by batch size, headline dictionary and a 1024 x 1024 one:
k segment-iterations/s, best of three rounds of ten encodes.   python scripts/fine_ab.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
for A, L, N, K in ((512, 512, 32768, 64), (1024, 1024, 32768, 64), (256, 256, 16384, 32)):
    dn = synth.make_dictionary(A, L, seed=1000)
    du = nat.unit_norm(torch.from_numpy(dn).cuda())
    for B in (1, 4, 8, 16, 24, 32, 48, 64, 128):
        x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=3 * K, seed=1002)).cuda()
        row = []
        for fine in (1, 2):
            nat.tune(nat.MP_TUNE_PERSIST_FINE, fine)
            for _ in range(4):
                nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_FFT_PERSISTENT)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10):
                    nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_FFT_PERSISTENT)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
            row.append(B * K / best / 1e3)
        print(f"{A} x {L}, B {B:3d}: one slot {row[0]:8.1f} k   two slots {row[1]:8.1f} k   {100 * (row[1] / row[0] - 1):+.1f} %", flush=True)
nat.tune(nat.MP_TUNE_PERSIST_FINE, 0)

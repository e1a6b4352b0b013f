#!/usr/bin/env python3
"""Host time of one asynchronous encode call (no synchronisation until the end): what a caller of many small encodes pays per
call whatever the GPU does.  python scripts/host_overhead.py"""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth, matchingpursuit as mp
A, L, N = 512, 512, 32768
dn = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(dn).cuda())
d = torch.from_numpy(dn).cuda()
for B, K in ((1, 2), (1, 8), (8, 8), (64, 64)):
    x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=3 * K, seed=1002)).cuda()
    for _ in range(5):
        nat.encode(x, du, K, path=nat.MP_PATH_FFT)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        nat.encode(x, du, K, path=nat.MP_PATH_FFT)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"B {B} K {K}: {t_issue / n * 1e6:.0f} us of host time per call to issue, {t_all / n * 1e6:.0f} us per call until the device is done", flush=True)
x = torch.from_numpy(synth.make_segments(1, N, dn, n_events=24, seed=1002)).cuda()
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    nat.encode(x, du, 8, path=nat.MP_PATH_FFT)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

"""Sub-batches at the BASELINE configs[3] shape (64 steps): none / 2 / 4 -- its screens fill the GPU alone, no gain
(245 / 269 / 246 ms), which is why the default leaves segments of >= 65536 cells on one stream."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "matching-pursuit_amd")
from mpcore import _native as nat, synth
A, L, N, B, K = 4096, 2048, 131072, 128, 64
d = synth.make_dictionary(A, L, seed=4000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=256, seed=4001)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
nat.encode(x[:4], du, 2, path=nat.MP_PATH_FFT); torch.cuda.synchronize()
for g in (1, 2, 4, 1):
    if g > 1: nat.tune(nat.MP_TUNE_GROUPS, g)
    flags = nat.MP_FLAG_NO_OVERLAP if g == 1 else nat.MP_FLAG_OVERLAP
    t0 = time.perf_counter(); nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"c4 shape B{B} K{K}, {g} sub-batch(es): {dt * 1e3:.1f} ms -> {B * K / dt:.0f} seg-it/s", flush=True)
nat.tune(nat.MP_TUNE_GROUPS, 4)

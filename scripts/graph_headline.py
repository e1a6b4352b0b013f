"""Headline shape: plain launches against the replay of a captured hipGraph (mpcore.EncodePlan), for the one-stream
schedule and the two-sub-batch default, with the sampled event profiling on and off."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "matching-pursuit_amd")
from mpcore import _native as nat, synth
A, L, N, B, K = 512, 512, 32768, 64, 64
d = synth.make_dictionary(A, L, seed=1000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
def rate(fn, n=8):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return B*K*n/(time.perf_counter()-t0)
for name, flags in (("one stream", nat.MP_FLAG_NO_OVERLAP), ("two sub-batches", 0)):
    for prof in (0, 16):
        nat.profile_enable(prof)
        plain = rate(lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags))
        nat.profile_read()
        plan = nat.EncodePlan(B, N, du, K, path=nat.MP_PATH_FFT, flags=flags)
        gr = rate(lambda: plan(x))
        nat.profile_read()
        print(f"{name:16s} profiling every {prof:2d}: plain {plain:9.0f}  graph replay {gr:9.0f}  ({gr/plain:.3f}x)", flush=True)
nat.profile_enable(0)

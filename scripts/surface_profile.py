import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, "matching-pursuit_amd")
from mpcore import matchingpursuit as mp, _native as nat, synth, encode_packed
A, L, N, B, K = 512, 512, 32768, 64, 64
d = torch.from_numpy(synth.make_dictionary(A, L, seed=1000)).cuda()
x = torch.from_numpy(synth.make_segments(B, N, synth.make_dictionary(A, L, seed=1000), n_events=192, seed=1002)).cuda()[:, None, :]
def surface():
    ev, scatter = mp.sparse_code(x, d, n_steps=K, flatten=True)
    return scatter(x.shape, ev)
for _ in range(30): surface()
torch.cuda.synchronize()
def T(f, n=30):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
print("surface", T(surface), "encode_packed", T(lambda: encode_packed(x, d, K)))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): surface()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)

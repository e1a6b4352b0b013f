#!/usr/bin/env python3
"""cProfile of dictionary_learning_step at the headline shape (where do the ~3 ms beside the encode go?)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import matchingpursuit as mp
from mpcore import synth
A, L, N, B, K = 512, 512, 32768, 64, 64
dn = synth.make_dictionary(A, L, seed=1000)
d = torch.from_numpy(dn).cuda()
x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=192, seed=1002)).cuda()[:, None, :]
for _ in range(3):
    mp.dictionary_learning_step(x, d, n_steps=K)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    mp.dictionary_learning_step(x, d, n_steps=K)
torch.cuda.synchronize()
print(f"dictionary_learning_step: {(time.perf_counter() - t0) * 100:.2f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    mp.dictionary_learning_step(x, d, n_steps=K)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

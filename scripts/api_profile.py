#!/usr/bin/env python3
"""cProfile of sparse_code(flatten=True) at the headline shape: where the host time of the reference-shaped
return values goes."""
import os, sys, cProfile, pstats
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import matchingpursuit as mp
from mpcore import synth
A, L, N, B, K = 512, 512, 32768, 64, 64
d = torch.from_numpy(synth.make_dictionary(A, L, seed=1000)).cuda()
x = torch.from_numpy(synth.make_segments(B, N, synth.make_dictionary(A, L, seed=1000), n_events=192, seed=1002)).cuda()[:, None, :]
for _ in range(3):
    mp.sparse_code(x, d, n_steps=K, flatten=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    mp.sparse_code(x, d, n_steps=K, flatten=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

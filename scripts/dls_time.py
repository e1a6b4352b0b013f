#!/usr/bin/env python3
"""Wall time of dictionary_learning_step (modules/matchingpursuit.py:348-419) at the headline shape."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import matchingpursuit as mp
from mpcore import synth
A, L, N, B, K = 512, 512, 32768, 64, 64
dn = synth.make_dictionary(A, L, seed=1000)
d = torch.from_numpy(dn).cuda()
x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=192, seed=1002)).cuda()[:, None, :]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d2 = mp.dictionary_learning_step(x, d, n_steps=K)
    torch.cuda.synchronize(); print(f"dictionary_learning_step: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
print("changed atoms:", int((d2 - torch.nn.functional.normalize(d, dim=-1)).abs().amax(dim=-1).gt(1e-6).sum()))

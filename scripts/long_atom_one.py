#!/usr/bin/env python3
"""One long-atom encode for the profiler: 1024 atoms of N/4 samples, B segments of N samples, 32 steps, planted events.
python3 scripts/long_atom_one.py [N] [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
A, L, K = 1024, N // 4, 32
d = synth.make_dictionary(A, L, seed=N)
du = nat.unit_norm(torch.from_numpy(d).cuda())
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=48, seed=N)).cuda()
for _ in range(2):
    out = nat.encode(x, du, K, path=nat.MP_PATH_FFT)
torch.cuda.synchronize()
print("done", int(torch.isnan(out[2]).any()), flush=True)

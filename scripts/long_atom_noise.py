#!/usr/bin/env python3
"""One long-atom encode of plain noise for the profiler (many contenders per step: the select side of the step).
python3 scripts/long_atom_noise.py [N] [batch] [flags]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
FLAGS = int(sys.argv[3]) if len(sys.argv) > 3 else 0
A, L, K = 1024, N // 4, 32
d = synth.make_dictionary(A, L, seed=N)
du = nat.unit_norm(torch.from_numpy(d).cuda())
x = torch.from_numpy(np.random.default_rng(N).standard_normal((B, N)).astype(np.float32)).cuda()
for _ in range(2):
    out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=FLAGS)
torch.cuda.synchronize()
print("done", int(torch.isnan(out[2]).any()), flush=True)

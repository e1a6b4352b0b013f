#!/usr/bin/env python3
"""A few persistent-schedule encodes at the headline shape (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth
A, L, N, K = 512, 512, 32768, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
for _ in range(4):
    nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_FFT_PERSISTENT)
    torch.cuda.synchronize()
print(nat.persist_stats())

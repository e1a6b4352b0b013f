#!/usr/bin/env python3
"""One warm encode per schedule, for a rocprofv3 --kernel-trace timeline (scripts/overlap_trace.py)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth
A, L, N, K, B = 512, 512, 32768, 64, 64
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 2
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
flags = nat.MP_FLAG_NO_OVERLAP if groups == 1 else (nat.MP_FLAG_OVERLAP | nat.flag_groups(groups))
nat.init_streams()
for _ in range(3):
    nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
    torch.cuda.synchronize()

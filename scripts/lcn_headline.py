#!/usr/bin/env python3
"""The local-contrast-norm schedule (mp_encode_lcn_f32) at the headline shape: 512 x 512 dictionary, 64 x 32768-sample segments,
K = 64 (argv: batch, steps).  Three encodes after one warm-up; for rocprofv3 --kernel-trace --stats / --pmc passes
(scripts/profile_round4.sh) and plain timing."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
A, L, N = 512, 512, 32768
d = synth.make_dictionary(A, L, seed=1000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=3 * K, seed=1002)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
nat.encode_lcn(x, du, K)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out = nat.encode_lcn(x, du, K)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"lcn {A}x{L}, {B} x {N}, K={K}: {dt * 1e3:.2f} ms per encode = {B * K / dt / 1e3:.1f} k segment-iterations/s", flush=True)

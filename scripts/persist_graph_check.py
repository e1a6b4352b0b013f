#!/usr/bin/env python3
"""The persistent form captured into a hipGraph against the plain call: which outputs differ, where."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth
A, L, N = 512, 512, 32768
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())


def run(tag, B, K, flags, warm=True):
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=3 * K, seed=1002)).cuda()
    ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_NO_OVERLAP)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    if warm:
        with torch.cuda.stream(side):
            nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
        torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
    sched = nat.last_schedule()
    for rep in range(3):
        g.replay()
        torch.cuda.synchronize()
        nan = int(torch.isnan(out[2]).any(dim=1).sum())
        same = all(torch.equal(p, q) for p, q in zip(out, ref))
        print(f"{tag} B{B} K{K} schedule {sched} replay {rep}: identical {same} nan rows {nan}", flush=True)


run("quarter-one-stream", 64, 8, nat.MP_FLAG_FFT_QUARTER | nat.MP_FLAG_NO_OVERLAP)
run("sub-batches", 64, 8, nat.MP_FLAG_FFT_NO_PERSISTENT)
run("persistent-small", 8, 4, nat.MP_FLAG_FFT_PERSISTENT)
run("persistent", 64, 64, 0)

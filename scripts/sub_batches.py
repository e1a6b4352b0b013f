"""How many sub-batches on forked streams?  Headline dictionary, one encode at a time (synchronised), batch sizes from
the command line (default 64 128 256).  The internal streams are chosen to be mutually concurrent (stream_pool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "matching-pursuit_amd")
from mpcore import _native as nat, synth
A, L, N, K = 512, 512, 32768, 64
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
for B in (tuple(int(v) for v in sys.argv[1:]) or (64, 128, 256)):
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
    row = []
    for g in (1, 2, 3, 4):
        if g > 1: nat.tune(nat.MP_TUNE_GROUPS, g)
        flags = nat.MP_FLAG_NO_OVERLAP if g == 1 else nat.MP_FLAG_OVERLAP
        f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
        f(); f(); torch.cuda.synchronize(); ts = []
        for _ in range(9):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        row.append(f"{g}: {B * K / float(np.median(ts)):8.0f}")
    print(f"B{B:4d} sub-batches " + "  ".join(row), flush=True)
nat.tune(nat.MP_TUNE_GROUPS, 2)

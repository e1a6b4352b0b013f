"""hipGraph replay of one encode (mpcore.EncodePlan) captured with 1 .. 4 sub-batches, headline dictionary."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "matching-pursuit_amd")
from mpcore import _native as nat, synth
A, L, N, K = 512, 512, 32768, 64
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
for B in (64, 128):
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
    row = []
    for g in (1, 2, 3, 4):
        flags = nat.MP_FLAG_NO_OVERLAP if g == 1 else nat.MP_FLAG_OVERLAP
        plan = nat.EncodePlan(B, N, du, K, path=nat.MP_PATH_FFT, flags=flags, sub_batches=max(g, 2))
        f = lambda: plan(x)
        f(); f(); torch.cuda.synchronize(); ts = []
        for _ in range(9):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        row.append(f"{g}: {B * K / float(np.median(ts)):8.0f}")
        del plan
    print(f"B{B:4d} graph replay, sub-batches " + "  ".join(row), flush=True)

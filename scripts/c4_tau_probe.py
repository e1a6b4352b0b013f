#!/usr/bin/env python3
"""What would a tighter screen bound buy at BASELINE configs[3]'s shape (4096 x 2048, 128 x 131072, K = 256, SURVEY 8(d)'s
planted signal)?  An experiment: mp_tune(MP_TUNE_TAU) sets a constant in place of the rigorous model (the audit has the model
used to 17 % at most, so a half and a quarter of it still hold on these inputs); the library default schedule, timed once
after one warm-up per setting.   python scripts/c4_tau_probe.py [segments]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
A, L, N, K = 4096, 2048, 131072, 256
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
d = synth.make_dictionary(A, L, seed=4000)
x = torch.empty(B, N, device="cuda")
for b0 in range(0, B, 32):
    x[b0:b0 + 32] = torch.from_numpy(synth.make_segments(min(32, B - b0), N, d, n_events=3 * K, seed=4001, first_index=b0)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
u = 2.0 ** -24
model = u * (1.001 * 0.58 * L + 4 * np.ceil(np.log2(4 * L)))
ref = None
for scale in (0.0, 0.5, 0.25):
    nat.tune(nat.MP_TUNE_TAU, scale * model)
    out = nat.encode(x, du, K, path=nat.MP_PATH_FFT)
    torch.cuda.synchronize(); nat.lazy_stats()
    t0 = time.perf_counter()
    out = nat.encode(x, du, K, path=nat.MP_PATH_FFT)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ls = nat.lazy_stats()
    if ref is None:
        ref = out
    same = all(torch.equal(p, q) for p, q in zip(out[:3], ref[:3]))
    print(f"tau {'model' if scale == 0 else f'{scale} x model'}: {dt * 1e3:.1f} ms = {B * K / dt / 1e3:.1f} k seg-it/s; "
          f"tile screens skipped {ls['skipped']} of {ls['decided']}; contender cells refined {ls['contender_cells']}; "
          f"marked {int(torch.isnan(out[2]).any(dim=1).sum())}; same events as the model's run: {same}", flush=True)
nat.tune(nat.MP_TUNE_TAU, 0)

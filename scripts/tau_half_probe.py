#!/usr/bin/env python3
"""What would a screen bound half as wide buy on signals that are NOT sparse in the dictionary?  (An experiment: mp_tune(MP_TUNE_TAU)
sets a constant in place of the rigorous bound; the audit says the bound is used to 17 % at most, so half of it still holds here.)
python scripts/tau_half_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
u = 2.0 ** -24
for A, L, N, B, K in ((512, 512, 32768, 64, 64), (1024, 4096, 16384, 8, 32), (1024, 8192, 32768, 8, 32), (1024, 2048, 8192, 8, 32)):
    dn = synth.make_dictionary(A, L, seed=L)
    du = nat.unit_norm(torch.from_numpy(dn).cuda())
    model = u * (1.001 * 0.58 * L + 4 * np.ceil(np.log2(2 * L)))
    rng = np.random.default_rng(L)
    for kind, xh in (("bed + noise, nothing planted", synth.make_segments(B, N, dn, n_events=0, seed=7)), ("white noise", rng.standard_normal((B, N)).astype(np.float32))):
        x = torch.from_numpy(xh).cuda()
        row = []
        for scale in (0.0, 0.5, 0.25):
            nat.tune(nat.MP_TUNE_TAU, scale * model)   # 0: the rigorous model
            for _ in range(3):
                out = nat.encode(x, du, K, path=nat.MP_PATH_FFT)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                out = nat.encode(x, du, K, path=nat.MP_PATH_FFT)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            row.append(f"{'model' if scale == 0 else f'{scale} x'}: {dt * 1e3:.2f} ms, {int(torch.isnan(out[2]).any(dim=1).sum())} marked")
        print(f"{A} x {L}, B {B}, {kind}: " + "; ".join(row), flush=True)
nat.tune(nat.MP_TUNE_TAU, 0)

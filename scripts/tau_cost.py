#!/usr/bin/env python3
"""What does the screen's error bound cost?  Encode time and overflow-marked segments with the library's bound
eps = u ||window|| max_a (1.001 W_a + 4 log2 M ||d_a||) (rigorous for the chain's rounding; W_a ~ 0.58 L) against
constant forms eps = tau ||window||: round 1's 2e-5, a sqrt(L)-type one, and the cruder rigorous (L + 4 log2 M) u,
at the headline shape, the config-4 shape and the multiband model's longest atoms."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth

def run(name, A, L, N, B, K, taus, reps=3):
    d = synth.make_dictionary(A, L, seed=1000)
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=min(3 * K, 192), seed=1002)).cuda()
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    ref = None
    for tname, tau in taus:
        nat.tune(nat.MP_TUNE_TAU, tau)
        out = nat.encode(x, du, K, path=nat.MP_PATH_FFT)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = nat.encode(x, du, K, path=nat.MP_PATH_FFT)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        marked = int(torch.isnan(out[2]).any(dim=1).sum())
        if ref is None:
            ref = out
        same = all(torch.equal(a, b) for a, b in zip(out, ref)) if marked == 0 else None
        print(f"{name:10s} tau {tname:22s} {dt * 1e3:9.3f} ms  {B * K / dt:12.0f} seg-it/s  marked {marked}  same events {same}", flush=True)
    nat.tune(nat.MP_TUNE_TAU, 0)

u = 2.0 ** -24
run("c2", 512, 512, 32768, 64, 64, [("library (~342 u)", 0), ("2e-5 (335 u)", 2e-5), ("sqrt-type 8(sqrtL+logM)u", 8 * (512 ** .5 + 11) * u), ("(L + 4 logM) u", (1.01 * 512 + 44) * u)], reps=5)
run("c4/8", 4096, 2048, 131072, 16, 64, [("library (~1250 u)", 0), ("2e-5 (335 u)", 2e-5), ("sqrt-type", 8 * (2048 ** .5 + 13) * u), ("(L + 4 logM) u", (1.01 * 2048 + 52) * u)], reps=2)
run("long", 256, 8192, 32768, 16, 32, [("library (~4840 u)", 0), ("2e-5 (335 u)", 2e-5), ("sqrt-type", 8 * (8192 ** .5 + 15) * u), ("(L + 4 logM) u", (1.01 * 8192 + 60) * u)], reps=2)
run("mid", 128, 128, 8192, 64, 32, [("library", 0), ("2e-5", 2e-5)], reps=5)

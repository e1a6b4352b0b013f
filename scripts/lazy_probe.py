#!/usr/bin/env python3
"""Lazy screen: events bit-identical with and without the coherence table, how many screen tasks it answers without a
transform, and what it is worth.  Usage: python scripts/lazy_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
if os.environ.get('LAZY_RADIUS'): nat.tune(nat.MP_TUNE_LAZY_RADIUS, int(os.environ['LAZY_RADIUS']))
A, L, N, K = 512, 512, 32768, 64
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
for exact in (True, False):
    nat.coherence_table(du, exact=exact); torch.cuda.synchronize()
    t1 = time.perf_counter(); mu_ = nat.coherence_table(du, exact=exact); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"coherence table ({'exact correlations' if exact else 'FFT screen'}) {tuple(mu_.shape)}: {(t2 - t1) * 1e3:.2f} ms; min {float(mu_.min()):.4f} "
          f"mean {float(mu_.mean()):.4f} max {float(mu_.max()):.4f}", flush=True)
    if exact:
        mu_exact = mu_
mu = mu_
print(f"screen table - exact table: min {float((mu - mu_exact).min()):.2e} max {float((mu - mu_exact).max()):.2e} (must be >= about -1e-4: both bound the same quantity)", flush=True)
margins = [float(v) for v in sys.argv[1:]] or [0.7]
for B in ([64, 128] if os.environ.get('LAZY_RADIUS') else [32, 48, 64, 96, 128]):
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
    ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_NO_OVERLAP)
    torch.cuda.synchronize()
    for name, co, mg in ([] if os.environ.get("LAZY_RADIUS") else [("plain", False, 1.0), ("auto", None, 0.7)]) + [(f"lazy {m:.2f}", mu, m) for m in margins]:
        nat.tune(nat.MP_TUNE_LAZY_MARGIN, mg)
        f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=co)
        out = f(); torch.cuda.synchronize()
        same = all(torch.equal(p, q) for p, q in zip(out, ref))
        t0 = time.perf_counter()
        for _ in range(8): out = f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
        st = nat.persist_stats()
        print(f"B{B:4d} {name:9s}: {dt * 1e3:6.2f} ms = {B * K / dt / 1e3:6.0f} k seg-it/s, identical {same}, tasks run {st['tasks']}, skipped {st['skipped']} "
              f"({100.0 * st['skipped'] / max(st['tasks'] + st['skipped'], 1):.1f} %), select {st['select_ticks'] / max(st['selects'], 1) / 100:.1f} us {st['select_phase_us']}, error {st['error']}", flush=True)

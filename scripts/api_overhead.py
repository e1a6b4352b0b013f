#!/usr/bin/env python3
"""What the reference-shaped return values cost on top of the encode: sparse_code (per-event Python tuples,
modules/matchingpursuit.py:229-345) against encode_packed, headline shape."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import matchingpursuit as mp
from mpcore import encode_packed, synth
A, L, N, B, K = 512, 512, 32768, 64, 64
d = torch.from_numpy(synth.make_dictionary(A, L, seed=1000)).cuda()
x = torch.from_numpy(synth.make_segments(B, N, synth.make_dictionary(A, L, seed=1000), n_events=192, seed=1002)).cuda()[:, None, :]
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); out = f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), out
t_packed, _ = timed(lambda: encode_packed(x, d, K))
t_inst, _ = timed(lambda: mp.sparse_code(x, d, n_steps=K))
t_flat, out = timed(lambda: mp.sparse_code(x, d, n_steps=K, flatten=True))
t_dec, _ = timed(lambda: out[1](x.shape, out[0]))
print(f"encode_packed {t_packed:.2f} ms | sparse_code (dict of event tuples) {t_inst:.2f} ms | flatten=True {t_flat:.2f} ms | "
      f"scatter_segments(shape, events) {t_dec:.2f} ms   [{B * K} events]")
